"""Parity of the hand-written gfx950 path with the CPU oracle, through the C ABI.

Bar (BASELINE.json north_star): restriction indices bit-exact; residual / Jacobian action
within 1e-10 relative on identical inputs.  Everything here calls the product library
(ceedpetscsolid_amd/csrc/libceed_mi355x.so); the oracle is only the checker."""
import os

import numpy as np
import pytest

from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import box_mesh, hollow_cylinder_mesh, load_mesh_npz
from ceedpetscsolid_amd.solid import SolidProblem, smooth_displacement
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10  # north_star tolerance for the floating-point path


def distorted_box(nx, ny, nz, seed=0, amp=0.04):
    m = box_mesh(nx, ny, nz)
    rng = np.random.default_rng(seed)
    m.coords += amp / max(nx, ny, nz) * rng.uniform(-1, 1, m.coords.shape)
    return m


def build_pair(oracle, gpu, mesh, degree, problem, **kw):
    a = SolidProblem(oracle, mesh, degree, problem, **kw)
    b = SolidProblem(gpu, mesh, degree, problem, **kw)
    return a, b


def vec_pair(pa, pb, n, arr=None):
    va, vb = pa.ceed.vector(n), pb.ceed.vector(n)
    if arr is not None:
        va.set_array(arr); vb.set_array(arr)
    return va, vb


CASES = [  # (mesh factory, degree, problem, bc)
    ("box p1", lambda: distorted_box(3, 2, 2), 1, "hyperFS", dict(bc_sides=[1])),
    ("box p2", lambda: distorted_box(4, 4, 4), 2, "linElas", dict(bc_all_boundary=True)),      # config 1 shape
    ("box p2 fs", lambda: distorted_box(3, 3, 2), 2, "hyperFS", dict(bc_sides=[1, 2])),
    ("box p3 ss", lambda: distorted_box(3, 2, 3), 3, "hyperSS", dict(bc_sides=[6])),
    ("cyl p4 fs", lambda: hollow_cylinder_mesh(2, 8, 3), 4, "hyperFS", dict(bc_sides=[998, 999])),
    ("cyl p4 ss", lambda: hollow_cylinder_mesh(2, 8, 3), 4, "hyperSS", dict(bc_sides=[998])),
    ("box p6 fs", lambda: distorted_box(2, 2, 2), 6, "hyperFS", dict(bc_sides=[1])),              # config 5 shape
    ("ragged", lambda: distorted_box(5, 1, 1), 4, "hyperFS", dict()),   # nelem not a multiple of the block's elements; no BC
    ("ragged p2", lambda: distorted_box(3, 1, 1), 2, "hyperSS", dict(bc_sides=[6])),   # Q=3: two elements per wave, 3 elements
    ("ragged p1", lambda: distorted_box(3, 1, 1), 1, "hyperFS", dict(bc_sides=[6])),   # Q=2: eight elements per wave, 3 elements
    ("single element", lambda: distorted_box(1, 1, 1), 3, "hyperFS", dict()),
    ("uniform ladder", lambda: distorted_box(2, 2, 2), 4, "linElas", dict(bc_sides=[1], multigrid="uniform")),  # P = 2,3,4,5 at Q = 5
    ("p5 ladder", lambda: distorted_box(2, 2, 1), 5, "hyperFS", dict(bc_sides=[1])),             # degrees 1,2,4,5: P = 2,3,5,6 at Q = 6
    ("p7 ladder", lambda: distorted_box(1, 2, 1), 7, "hyperSS", dict(bc_sides=[1])),             # degrees 1,2,4,7: P = 2,3,5,8 at Q = 8
    ("p5 uniform", lambda: distorted_box(2, 1, 1), 5, "linElas", dict(bc_sides=[1], multigrid="uniform")),  # P = 2..6 at Q = 6
    ("p6 uniform", lambda: distorted_box(1, 1, 2), 6, "hyperFS", dict(bc_sides=[1], multigrid="uniform")),  # P = 2..7 at Q = 7
    ("p7 uniform", lambda: distorted_box(1, 1, 1), 7, "hyperSS", dict(bc_sides=[1], multigrid="uniform")),  # P = 2..8 at Q = 8
    # -qextra 1, 2 (src/cloptions.c:53-55): Q = degree + 1 + qextra on EVERY level -- the residual kernel with its stored state, the
    # Jacobian and the diagonal of the FINE level at P < Q
    ("p2 qextra1 le", lambda: distorted_box(3, 2, 2), 2, "linElas", dict(bc_sides=[1], qextra=1)),      # P = 2, 3 at Q = 4
    ("p2 qextra1 ss", lambda: distorted_box(2, 3, 2), 2, "hyperSS", dict(bc_sides=[6], qextra=1)),
    ("p2 qextra2 fs", lambda: distorted_box(3, 2, 2), 2, "hyperFS", dict(bc_sides=[1, 2], qextra=2)),   # P = 2, 3 at Q = 5
    ("p3 qextra1 fs", lambda: hollow_cylinder_mesh(2, 8, 2), 3, "hyperFS", dict(bc_sides=[998], qextra=1)),   # P = 2, 3, 4 at Q = 5 (swept)
    ("p3 qextra2 ss", lambda: distorted_box(2, 2, 2), 3, "hyperSS", dict(bc_sides=[1], qextra=2)),      # P = 2, 3, 4 at Q = 6
    ("p3 qextra2 le", lambda: box_mesh(2, 2, 3), 3, "linElas", dict(bc_sides=[1], qextra=2)),           # affine elements at Q = 6
]


@pytest.mark.parametrize("name,mk,degree,problem,bc", CASES, ids=[c[0] for c in CASES])
def test_all_operators_match_oracle(oracle, gpu, name, mk, degree, problem, bc):
    mesh = mk()
    pa, pb = build_pair(oracle, gpu, mesh, degree, problem, nu=0.3, E=2.5, **bc)
    rng = np.random.default_rng(42)
    # geometry (opSetupGeo): identical layout [elem][comp][point] on both backends
    assert rel_err(pb.qdata.to_numpy(), pa.qdata.to_numpy()) < 1e-13
    n = pa.lsize()
    u = pa.smooth_state(0.15)
    # residual (+ stored state)
    xa, xb = vec_pair(pa, pb, n, u)
    ya, yb = vec_pair(pa, pb, n)
    pa.form_residual(xa, ya); pb.form_residual(xb, yb)
    assert "fused_grad" in pb.opApply.kernel_name
    assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL
    if pa.gradu is not None:
        assert rel_err(pb.gradu.to_numpy(), pa.gradu.to_numpy()) < 1e-12
    # Jacobian, transfer and diagonal on every level
    for lv in range(len(pa.levels)):
        nl = pa.lsize(lv)
        x = rng.uniform(-1, 1, nl)
        xa, xb = vec_pair(pa, pb, nl, x)
        ya, yb = vec_pair(pa, pb, nl)
        pa.apply_jacobian(lv, xa, ya); pb.apply_jacobian(lv, xb, yb)
        ja, jb = ya.to_numpy(), yb.to_numpy()
        assert rel_err(jb, ja) < TOL, (lv, rel_err(jb, ja))
        assert np.all(jb[pa.levels[lv].mask != 0] == 0.0)       # constrained rows dropped
        da, db = vec_pair(pa, pb, nl)
        db.set_value(7.0)                                        # overwrite semantics (matops.c:227)
        pa.get_diag(lv, da); pb.get_diag(lv, db)
        assert rel_err(db.to_numpy(), da.to_numpy()) < TOL
        assert rel_err(pb.levels[lv].multinv.to_numpy(), pa.levels[lv].multinv.to_numpy()) == 0.0
        if lv > 0:
            nc = pa.lsize(lv - 1)
            xc = rng.uniform(-1, 1, nc)
            ca, cb = vec_pair(pa, pb, nc, xc)
            fa, fb = vec_pair(pa, pb, nl)
            pa.prolong(lv, ca, fa); pb.prolong(lv, cb, fb)
            assert rel_err(fb.to_numpy(), fa.to_numpy()) < TOL
            fa, fb = vec_pair(pa, pb, nl, x)
            ca, cb = vec_pair(pa, pb, nc)
            pa.restrict(lv, fa, ca); pb.restrict(lv, fb, cb)
            assert rel_err(cb.to_numpy(), ca.to_numpy()) < TOL


@pytest.mark.parametrize("name", __import__("conftest").operator_golden_cases())
def test_operators_match_reference_callbacks(gpu, name):
    """The HIP path against vectors the REFERENCE's own compiled callbacks produced inside whole operators
    (tests/golden/operators.npz, oracle/gen_operator_golden.py: hyperFS.h:147-464, hyperSS.h:60-321, linElas.h:39-280,
    common.h:47-101 as object code): residual, stored state, Jacobian action and diagonal on every level at the parity bar;
    meshes with general, swept and affine elements so that every geometry form of the fused kernel is compared."""
    from conftest import check_against_operator_golden, operator_golden_problem
    p, f = operator_golden_problem(gpu, name)
    check_against_operator_golden(p, f, TOL)
    want = {"affine": "affine elements", "swept": "swept elements", "general": "recomputed"}
    for key, text in want.items():
        if name.endswith(key):
            assert text in p.levels[p.fine].opJacob.kernel_name, p.levels[p.fine].opJacob.kernel_name
    p.destroy()


def test_restriction_indices_bit_exact(oracle, gpu):
    """Gather through the offsets is a pure copy: the E-vectors must be bitwise identical."""
    mesh = distorted_box(3, 3, 2)
    from ceedpetscsolid_amd.mesh import build_dofmap
    dm = build_dofmap(mesh, 3)
    rng = np.random.default_rng(5)
    lv = rng.uniform(-1, 1, dm.lsize)
    outs = []
    for c in (oracle, gpu):
        r = c.elem_restriction(mesh.nelem, dm.P ** 3, 3, 1, dm.lsize, dm.offsets())
        L = c.vector(dm.lsize).set_array(lv)
        E = r.create_evector()
        r.apply(cd.NOTRANSPOSE, L, E)
        M = r.create_lvector(); r.multiplicity(M)
        outs.append((E.to_numpy(), M.to_numpy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[1][1].sum() == mesh.nelem * dm.P ** 3 * 3


def test_log1p_range_shift_branches_on_gpu(oracle, gpu):
    """hyperFS.h:49-55: drive det(C)-1 outside (sqrt2/2-1, sqrt2-1) with a large uniform
    stretch / compression so both range-shift branches of the series run on the GPU."""
    mesh = box_mesh(2, 2, 2)
    for scale in (-0.25, 0.3):
        pa, pb = build_pair(oracle, gpu, mesh, 2, "hyperFS", nu=0.3, E=1.0)
        n = pa.lsize()
        u = (scale * pa.levels[pa.fine].dofmap.node_coords).reshape(-1)
        xa, xb = vec_pair(pa, pb, n, u); ya, yb = vec_pair(pa, pb, n)
        pa.form_residual(xa, ya); pb.form_residual(xb, yb)
        assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL
        x = np.random.default_rng(3).uniform(-1, 1, n)
        xa, xb = vec_pair(pa, pb, n, x)
        pa.apply_jacobian(pa.fine, xa, ya); pb.apply_jacobian(pb.fine, xb, yb)
        assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL


@pytest.mark.parametrize("nu,amp", [(0.49, 0.15), (0.3, 1e-7), (0.3, 0.3), (0.45, 0.3)],
                         ids=["nearly incompressible", "tiny strain", "large strain", "large strain nu .45"])
def test_hyperfs_tangent_spatial_form_extremes(oracle, gpu, nu, amp):
    """The device evaluates HyperFSdF (hyperFS.h:286-464) in the algebraically equal spatial form
    dP = mu grad(du) + (lambda tr(h) I - f h^T) F^-T (qfunctions_device.hpp); the oracle keeps the reference's
    order of operations.  The two must agree to the parity bar where rounding differs most: lambda >> mu,
    strains at rounding level, and strains of 30 % on distorted elements."""
    mesh = distorted_box(3, 2, 2, seed=4, amp=0.15)
    pa, pb = build_pair(oracle, gpu, mesh, 3, "hyperFS", nu=nu, E=3.0, bc_sides=[1])
    n = pa.lsize()
    u = pa.smooth_state(amp)
    xa, xb = vec_pair(pa, pb, n, u); ya, yb = vec_pair(pa, pb, n)
    pa.form_residual(xa, ya); pb.form_residual(xb, yb)
    x = np.random.default_rng(11).uniform(-1, 1, n)
    xa, xb = vec_pair(pa, pb, n, x)
    pa.apply_jacobian(pa.fine, xa, ya); pb.apply_jacobian(pb.fine, xb, yb)
    assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL
    da, db = vec_pair(pa, pb, n)
    pa.get_diag(pa.fine, da); pb.get_diag(pb.fine, db)
    assert rel_err(db.to_numpy(), da.to_numpy()) < TOL


def test_config2_cube4096_p3_linelas(oracle, gpu):
    """BASELINE config 2: linElas, cube8_4096e_6ss_s, degree 3, Jacobian apply vs the CPU path."""
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cube8_4096e_6ss_s.npz"))
    pa, pb = build_pair(oracle, gpu, mesh, 3, "linElas", nu=0.3, E=1e6, bc_sides=[999] if 999 in mesh.side_sets else [992])
    n = pa.lsize()
    assert n == 352947
    x = np.random.default_rng(0).uniform(-1, 1, n)
    xa, xb = vec_pair(pa, pb, n, x); ya, yb = vec_pair(pa, pb, n)
    pa.apply_jacobian(pa.fine, xa, ya); pb.apply_jacobian(pb.fine, xb, yb)
    assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL


def test_config3_cylinder5580_p4_hyperss(oracle, gpu):
    """BASELINE config 3's operator: hyperSS, cylinder8_5580e_4ss_us, degree 4 (levels 1,2,4)."""
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_5580e_4ss_us.npz"))
    pa, pb = build_pair(oracle, gpu, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
    n = pa.lsize()
    assert n == 1159692 and pa.degrees == [1, 2, 4]
    u = pa.smooth_state(0.05)
    xa, xb = vec_pair(pa, pb, n, u); ya, yb = vec_pair(pa, pb, n)
    pa.form_residual(xa, ya); pb.form_residual(xb, yb)
    assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL
    for lv in range(3):
        nl = pa.lsize(lv)
        x = np.random.default_rng(lv).uniform(-1, 1, nl)
        xa, xb = vec_pair(pa, pb, nl, x); ya, yb = vec_pair(pa, pb, nl)
        pa.apply_jacobian(lv, xa, ya); pb.apply_jacobian(lv, xb, yb)
        assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL


def test_full_size_properties_config4(gpu):
    """BASELINE config 4 at full size (hyperFS, ~99k-element hollow cylinder, degree 4): too big
    for the oracle in seconds, so size-independent properties: symmetry of the tangent
    (v'Jw = w'Jv), linearity, rigid-translation null space without BCs."""
    mesh = hollow_cylinder_mesh(10, 110, 90)
    assert mesh.nelem == 99000
    p = SolidProblem(gpu, mesh, 4, "hyperFS", nu=0.3, E=1.0, multigrid="none")
    n = p.lsize()
    c = p.ceed
    u = p.smooth_state(0.1)
    X, Y = c.vector(n).set_array(u), c.vector(n)
    p.form_residual(X, Y)
    rng = np.random.default_rng(11)
    v, w = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    def J(z):
        X.set_array(z); p.apply_jacobian(p.fine, X, Y); return Y.to_numpy()
    jv, jw = J(v), J(w)
    assert abs(v @ jw - w @ jv) < 1e-11 * abs(v @ jw)
    assert rel_err(J(2.0 * v - 0.5 * w), 2.0 * jv - 0.5 * jw) < 1e-12
    t = np.tile([0.3, -1.0, 2.0], n // 3)
    assert np.abs(J(t)).max() < 1e-11 * np.abs(jv).max()


@pytest.mark.parametrize("which", ["config 4", "config 5 per-GPU block", "unstructured cylinder8_44928e", "unstructured 99 216 hexes"])
def test_full_size_matches_oracle(oracle, oracle_lib, gpu, which):
    """BASELINE configs 4 and 5 (one GPU's 32^3 block, p=6) at FULL size against the oracle itself (threaded over
    elements: a residual and a Jacobian apply of a ~20 M-dof problem take the CPU a few seconds each): the north-star
    tolerance on the headline workloads, not only their size-independent properties."""
    import ctypes as C
    if which == "config 4":
        mesh, degree, bc = hollow_cylinder_mesh(10, 110, 90), 4, [998, 999]
    elif which == "unstructured 99 216 hexes":   # config 4's SIZE on the reference's own CUBIT-paved cross-section (468 quads, refined 2 x 2, 53 layers)
        from ceedpetscsolid_amd.mesh import refine_swept_mesh
        mesh, degree, bc = refine_swept_mesh(load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_44928e_2ss_us.npz")), 53), 4, [998, 999]
        assert mesh.nelem == 99216
    elif which.startswith("unstructured"):   # the largest unstructured reference cylinder present: CUBIT element and vertex order
        mesh, degree, bc = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_44928e_2ss_us.npz")), 4, [998, 999]
        assert mesh.nelem == 44928
    else:
        mesh, degree, bc = box_mesh(32, 32, 32), 6, [1, 2]
    nthreads = max(1, min(16, len(os.sched_getaffinity(0))))
    oracle_lib.lib.OracleSetNumThreads(C.c_int(nthreads))
    try:
        pa, pb = build_pair(oracle, gpu, mesh, degree, "hyperFS", nu=0.3, E=1.0, bc_sides=bc, multigrid="none")
        n = pa.lsize()
        u = pa.smooth_state(0.1)
        xa, xb = vec_pair(pa, pb, n, u); ya, yb = vec_pair(pa, pb, n)
        pa.form_residual(xa, ya); pb.form_residual(xb, yb)
        assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL
        x = np.random.default_rng(4).uniform(-1, 1, n)
        xa, xb = vec_pair(pa, pb, n, x)
        pa.apply_jacobian(pa.fine, xa, ya); pb.apply_jacobian(pb.fine, xb, yb)
        assert "/pencil" in pb.levels[pb.fine].opJacob.kernel_name
        assert rel_err(yb.to_numpy(), ya.to_numpy()) < TOL
    finally:
        oracle_lib.lib.OracleSetNumThreads(C.c_int(1))


def test_full_size_transfers_and_fused_step_config4(oracle, oracle_lib, gpu):
    """Round 5 at BASELINE config 4's FULL size (99 000 hexes, levels p = 1, 2, 4): Prolong_Ceed / Restrict_Ceed in owner form against
    the oracle's sum over the sharing elements at 1e-13 on both level pairs (49 500 groups over the eight XCD chunks), Restrict =
    Prolong^T, and the apply fused with its consumer bitwise equal to the two-pass form on the fine level (19.4 M dofs)."""
    import ctypes as C
    mesh = hollow_cylinder_mesh(10, 110, 90)
    nthreads = max(1, min(16, len(os.sched_getaffinity(0))))
    oracle_lib.lib.OracleSetNumThreads(C.c_int(nthreads))
    try:
        pa, pb = build_pair(oracle, gpu, mesh, 4, "linElas", nu=0.3, E=1.0, bc_sides=[998, 999])
        rng = np.random.default_rng(8)
        for lv in (1, 2):
            nf, nc = pa.lsize(lv), pa.lsize(lv - 1)
            xc, xf = rng.uniform(-1, 1, nc), rng.uniform(-1, 1, nf)
            res = []
            for p in (pa, pb):
                Xc, Xf, Yf, Yc = p.ceed.vector(nc).set_array(xc), p.ceed.vector(nf).set_array(xf), p.ceed.vector(nf), p.ceed.vector(nc)
                p.prolong(lv, Xc, Yf); p.restrict(lv, Xf, Yc)
                res.append((Yf.to_numpy(), Yc.to_numpy()))
            assert rel_err(res[1][0], res[0][0]) < 1e-13 and rel_err(res[1][1], res[0][1]) < 1e-13, lv
            lhs, rhs = float(res[1][0] @ xf), float(xc @ res[1][1])
            assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), abs(rhs)), (lv, lhs, rhs)
        # one smoothing step on the fine level, fused and in two passes: same bits
        L, op, n = gpu.L, pb.levels[pb.fine].opJacob, pb.lsize()
        free = (pb.levels[pb.fine].mask == 0).astype(np.float64)
        a = {k: rng.uniform(-1, 1, n) * free for k in ("x", "d", "b")}
        a["dinv"] = rng.uniform(0.5, 2.0, n) * free
        out = []
        for fused in (False, True):
            v = {k: gpu.vector(n).set_array(a[k]) for k in a}
            t = gpu.vector(n)
            c1, c2 = C.c_double(0.3), C.c_double(0.2)
            if fused:
                L.chk(L.lib.CeedXOperatorApplyChebyshev(op.h, v["x"].h, t.h, v["x"].h, v["d"].h, None, v["b"].h, v["dinv"].h, c1, c2, 0))
            else:
                op.apply(v["x"], t)
                L.chk(L.lib.CeedXVectorChebyshevStep(v["x"].h, v["d"].h, None, v["b"].h, t.h, v["dinv"].h, c1, c2, 0))
            out.append((v["x"].to_numpy(), v["d"].to_numpy()))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    finally:
        oracle_lib.lib.OracleSetNumThreads(C.c_int(1))


def test_device_pointer_use_pointer_roundtrip(gpu):
    """matops.c:40-50 with -memtype device: SetArray(DEVICE, USE_POINTER) / TakeArray on
    buffers owned by the caller (torch tensors standing in for PETSc's device Vecs)."""
    import torch
    mesh = box_mesh(2, 2, 2)
    p = SolidProblem(gpu, mesh, 2, "linElas", nu=0.3, E=1.0, bc_sides=[1])
    n = p.lsize()
    x = torch.rand(n, dtype=torch.float64, device="cuda")
    y = torch.full((n,), 3.0, dtype=torch.float64, device="cuda")
    lv = p.levels[p.fine]
    lv.xceed.set_device_pointer(x.data_ptr()); lv.yceed.set_device_pointer(y.data_ptr())
    p.apply_jacobian(p.fine, lv.xceed, lv.yceed)
    lv.xceed.take_array(cd.MEM_DEVICE); lv.yceed.take_array(cd.MEM_DEVICE)
    torch.cuda.synchronize()
    X, Y = p.ceed.vector(n).set_array(x.cpu().numpy()), p.ceed.vector(n)
    p.apply_jacobian(p.fine, X, Y)
    assert rel_err(y.cpu().numpy(), Y.to_numpy()) < 1e-13


@pytest.mark.parametrize("memtype", ["device", "host"])
def test_extension_free_call_sequence_of_matops(oracle, gpu, memtype):
    """The UNCHANGED src/matops.c above this library: no CeedX* call at all.  ApplyLocalCeedOp (matops.c:26-60) borrows the
    caller's buffers with CeedVectorSetArray(memtype, CEED_USE_POINTER), applies, and takes them back; the caller zeroes the
    constrained entries of the input (G->L into a zeroed Xloc, :33,106), drops the constrained rows of the output (L->G,
    :57) and multiplies by multVec around the transfers (:149,176).  Residual, Jacobian / diagonal / prolong / restrict on
    every level, -memtype device (borrowed device buffers) and -memtype host (borrowed host buffers)."""
    import torch
    mesh = hollow_cylinder_mesh(2, 8, 3)
    kw = dict(nu=0.3, E=2.5, bc_sides=[998, 999])
    pa = SolidProblem(oracle, mesh, 4, "hyperFS", fused_bc=False, **kw)
    pb = SolidProblem(gpu, mesh, 4, "hyperFS", fused_bc=False, **kw)
    pf = SolidProblem(gpu, mesh, 4, "hyperFS", fused_bc=True, **kw)       # the fused form, for cross-checking
    rng = np.random.default_rng(21)

    def apply_local(p, op_apply, xin, nout):
        """ApplyLocalCeedOp: borrow, apply, take back.  xin: numpy L-vector; returns numpy."""
        c = p.ceed
        X, Y = c.vector(xin.size), c.vector(nout)
        if c is oracle or memtype == "host":
            xbuf, ybuf = np.ascontiguousarray(xin, dtype=np.float64), np.full(nout, 9.0)
            X.set_array(xbuf, copy=False); Y.set_array(ybuf, copy=False)           # CEED_MEM_HOST, CEED_USE_POINTER
            op_apply(X, Y)
            X.take_array(cd.MEM_HOST); Y.take_array(cd.MEM_HOST)
            return ybuf.copy()
        xt, yt = torch.from_numpy(xin).cuda(), torch.full((nout,), 9.0, dtype=torch.float64, device="cuda")
        X.set_device_pointer(xt.data_ptr()); Y.set_device_pointer(yt.data_ptr())   # CEED_MEM_DEVICE, CEED_USE_POINTER
        op_apply(X, Y)
        X.take_array(cd.MEM_DEVICE); Y.take_array(cd.MEM_DEVICE)
        torch.cuda.synchronize()
        return yt.cpu().numpy()

    fine = pa.fine
    free = [(lv.mask == 0).astype(np.float64) for lv in pa.levels]
    # FormResidual_Ceed (matops.c:63-79): boundary values stay in Xloc, constrained rows dropped on the way back
    u = pa.smooth_state(0.12)
    ra, rb = (apply_local(p, p.opApply.apply, u, u.size) * free[fine] for p in (pa, pb))
    assert rel_err(rb, ra) < TOL
    X, Y = gpu.vector(u.size).set_array(u), gpu.vector(u.size)
    pf.form_residual(X, Y)
    assert rel_err(rb, Y.to_numpy()) < 1e-13
    for lv in range(len(pa.levels)):
        n = pa.lsize(lv)
        x = rng.uniform(-1, 1, n)
        # ApplyJacobian_Ceed (matops.c:98-112)
        ja, jb = (apply_local(p, p.levels[lv].opJacob.apply, x * free[lv], n) * free[lv] for p in (pa, pb))
        assert rel_err(jb, ja) < TOL, lv
        X, Y = gpu.vector(n).set_array(x), gpu.vector(n)
        pf.apply_jacobian(lv, X, Y)
        assert rel_err(jb, Y.to_numpy()) < 1e-13, lv
        # GetDiag_Ceed (matops.c:206-244): the diagonal lands in the borrowed Yloc
        da, db = (apply_local(p, lambda X_, Y_, p=p: p.levels[lv].opJacob.assemble_diagonal(Y_), x, n) * free[lv] for p in (pa, pb))
        assert rel_err(db, da) < TOL, lv
        if lv == 0:
            continue
        nc = pa.lsize(lv - 1)
        xc = rng.uniform(-1, 1, nc)
        mult = pa.levels[lv].multinv.to_numpy()
        assert np.array_equal(mult, pb.levels[lv].multinv.to_numpy())
        # Prolong_Ceed (matops.c:115-157): apply, then VecPointwiseMult with multVec
        fa, fb = (apply_local(p, p.levels[lv].opProlong.apply, xc * free[lv - 1], n) * mult * free[lv] for p in (pa, pb))
        assert rel_err(fb, fa) < TOL, lv
        Xc, Yf = gpu.vector(nc).set_array(xc), gpu.vector(n)
        pf.prolong(lv, Xc, Yf)
        assert rel_err(fb, Yf.to_numpy()) < 1e-13, lv
        # Restrict_Ceed (matops.c:160-203): VecPointwiseMult with multVec, then apply
        ca, cb = (apply_local(p, p.levels[lv].opRestrict.apply, x * free[lv] * mult, nc) * free[lv - 1] for p in (pa, pb))
        assert rel_err(cb, ca) < TOL, lv
        Xf, Yc = gpu.vector(n).set_array(x), gpu.vector(nc)
        pf.restrict(lv, Xf, Yc)
        assert rel_err(cb, Yc.to_numpy()) < 1e-13, lv


@pytest.mark.parametrize("workload", ["cylinder p4", "box p6", "box p2"])
def test_one_call_apply_with_halo_equals_its_parts_bitwise(product_lib, workload):
    """CeedXOperatorApplyWithHalo (ApplyLocalCeedOp + DMLocalToGlobal(ADD_VALUES), matops.c:46,57, as one library call) on
    an EMULATED rank of a partitioned job: the rank's real sub-mesh and neighbour lists, the exchange sent to the rank
    itself through a one-rank RCCL communicator.  Three routes must agree BIT FOR BIT: (a) the one call -- two launches of
    the fused kernel on two streams, the exchange started behind the interface rows, the arrivals added by the launch
    that sums the interior rows; (b) round 2's sequence phase 0 / CeedXHaloStart / phase 1 / CeedXHaloFinish; (c) the whole
    apply followed by the exchange.  Recorded into a hipGraph and replayed on new data where this RCCL allows it."""
    from ceedpetscsolid_amd.halo import HaloExchange, RcclHalo, interface_elements, part_box, part_cylinder, virtual_world
    from ceedpetscsolid_amd.harness import SolidApp
    from ceedpetscsolid_amd.mesh import reorder_elements_first
    if workload == "cylinder p4":
        K, N, degree, part = 1, 4, 4, (lambda r: part_cylinder(r, 4, 3, 16, 16))
    elif workload == "box p6":
        K, N, degree, part = 3, 8, 6, (lambda r: part_box(r, 8, 6, 6, 6))
    else:
        K, N, degree, part = 5, 8, 2, (lambda r: part_box(r, 8, 12, 12, 12))
    mesh = part(K)
    vw = virtual_world(K, N, mesh, part, degree)
    lead = interface_elements(mesh, virtual=vw)
    assert lead.any() and not lead.all()
    mesh = reorder_elements_first(mesh, lead)
    bc = [s for s in (998, 999, 1, 2) if s in mesh.side_sets]
    want = None
    # every form of the one call: whole apply then exchange (0, default), split-phase on one stream (1), on two (2); the
    # RCCL group in order on the producing stream (default) or on the communicator's own stream
    for mode, inline in ((0, 1), (2, 1), (1, 1), (2, 0), (1, 0), (0, 0)):
        os.environ["CEED_MI355X_COMM_INLINE"] = str(inline)
        try:
            ceed = _ceed_with_env(product_lib, "CEED_MI355X_OVL_MODE", str(mode))
        finally:
            os.environ.pop("CEED_MI355X_COMM_INLINE", None)
        got = _one_call_routes(ceed, mesh, vw, lead, degree, bc, record=(True if inline else "refused") if mode in (0, 2) else False)
        if want is None:
            want = got
        assert np.array_equal(got, want), (mode, inline)


def _one_call_routes(ceed, mesh, vw, lead, degree, bc, record):
    from ceedpetscsolid_amd.halo import HaloExchange, RcclHalo
    from ceedpetscsolid_amd.harness import SolidApp
    app = SolidApp(ceed, mesh, degree, "hyperFS", nu=0.3, E=1.0, bc_sides=bc, multigrid="none")
    dm = app.dofmaps[app.fine]
    halo = HaloExchange(mesh, dm, device="cuda", virtual=vw)
    assert len(halo.neigh) >= 2
    ch = RcclHalo(ceed, halo, emulate_self=True)
    n = app.lsize()
    rng = np.random.default_rng(5)
    X, Ya, Yb, Yc = ceed.vector(n), ceed.vector(n), ceed.vector(n), ceed.vector(n)
    X.set_array(smooth_displacement(dm.node_coords, 0.05)); app.form_residual(X, Ya)
    op = app.opJacob[app.fine]
    op.set_overlap_split(int(lead.sum()), halo.interface_dof_mask())
    last = None
    for it in range(4):
        X.set_array(rng.uniform(-1, 1, n) * (app.masks[app.fine] == 0))
        Ya.set_value(7.0); Yb.set_value(-3.0); Yc.set_value(1.0)
        op.apply_with_halo(X, Ya, ch)                                     # (a)
        op.apply_phase(X, Yb, 0); ch.start(Yb); op.apply_phase(X, Yb, 1); ch.finish(Yb)    # (b)
        app.apply_jacobian(app.fine, X, Yc); ch.add(Yc)                   # (c)
        ya = Ya.to_numpy()
        assert np.isfinite(ya).all() and np.abs(ya).max() > 0
        assert np.array_equal(ya, Yb.to_numpy()), it
        assert np.array_equal(ya, Yc.to_numpy()), it
        last = ya
    # the C++ harness: ApplyJacobian_Ceed with the level's halo attached takes the one-call route
    app.set_halo(app.fine, ch)
    app.apply_jacobian(app.fine, X, Yb)
    assert np.array_equal(Yb.to_numpy(), Ya.to_numpy())
    app.set_halo(app.fine, None)
    Yg = ceed.vector(n)
    if record == "refused":   # RCCL on a stream of its own crashes inside a capture (tools/rccl_capture_probe.py): an error, not a crash
        with pytest.raises(cd.CeedError):
            ceed.capture(lambda: op.apply_with_halo(X, Yg, ch))
        op.apply_with_halo(X, Yg, ch)                     # and the Ceed is usable afterwards
        op.apply_with_halo(X, Ya, ch)
        assert np.array_equal(Yg.to_numpy(), Ya.to_numpy())
    elif record:              # in order on the capturing stream the exchange records and replays
        op.apply_with_halo(X, Yg, ch)
        g = ceed.capture(lambda: op.apply_with_halo(X, Yg, ch))
        for _ in range(3):
            X.set_array(rng.uniform(-1, 1, n) * (app.masks[app.fine] == 0)); X.device_pointer()
            op.apply_with_halo(X, Ya, ch)
            Yg.set_value(-7.0)
            g.launch()
            assert np.array_equal(Yg.to_numpy(), Ya.to_numpy())
        g.destroy()
    ch.destroy()
    return last


def test_split_phase_exchange_refuses_a_halo_outside_the_priority_rows(product_lib):
    """The split-phase one-call forms (CEED_MI355X_OVL_MODE 1 / 2) add the arrivals in the launch that overwrites the non-priority
    rows and start the exchange when only the priority rows are complete: a halo entry on a node outside the operator's priority
    set would be exchanged half-summed and overwritten.  Checked once per (operator, halo): an error, not a wrong sum."""
    import ctypes as C
    from ceedpetscsolid_amd.halo import HaloExchange, RcclHalo, interface_elements, part_cylinder, virtual_world
    from ceedpetscsolid_amd.harness import SolidApp
    from ceedpetscsolid_amd.mesh import reorder_elements_first
    ceed = _ceed_with_env(product_lib, "CEED_MI355X_OVL_MODE", "1")
    part = lambda r: part_cylinder(r, 4, 3, 12, 8)
    mesh = part(1)
    vw = virtual_world(1, 4, mesh, part, 2)
    lead = interface_elements(mesh, virtual=vw)
    mesh = reorder_elements_first(mesh, lead)
    app = SolidApp(ceed, mesh, 2, "hyperFS", nu=0.3, E=1.0, bc_sides=[], multigrid="none")
    dm = app.dofmaps[app.fine]
    halo = HaloExchange(mesh, dm, device="cuda", virtual=vw)
    good = RcclHalo(ceed, halo, emulate_self=True)
    n = app.lsize()
    X, Y = ceed.vector(n), ceed.vector(n)
    X.set_array(smooth_displacement(dm.node_coords, 0.05)); app.form_residual(X, Y)
    op = app.opJacob[app.fine]
    op.set_overlap_split(int(lead.sum()), halo.interface_dof_mask())
    op.apply_with_halo(X, Y, good)                      # the matching halo passes the check
    # a halo with one entry on an interior (non-priority) node
    prio = halo.interface_dof_mask()
    free = np.flatnonzero(prio == 0); stray = int(free[len(free) // 2])
    idx = np.array([stray], dtype=np.int32)
    h = C.c_void_p()
    ceed.L.chk(ceed.L.lib.CeedXHaloCreate(ceed.h, 1, (C.c_int * 1)(0), (C.c_int * 1)(1), (C.POINTER(C.c_int) * 1)(idx.ctypes.data_as(C.POINTER(C.c_int))), C.byref(h)))
    with pytest.raises(cd.CeedError, match="not on a priority node"):
        op.apply_with_halo(X, Y, h)
    op.apply_with_halo(X, Y, good)                      # still usable
    ceed.L.chk(ceed.L.lib.CeedXHaloDestroy(C.byref(h)))
    good.destroy()


def test_recorded_graph_refuses_to_replay_on_dropped_provenance(product_lib):
    """A recorded apply holds the provenance buffers of its passive vectors in its kernel arguments: the derived state beside grad u
    (HyperFSdF from Q = 6 on) and the element-map coefficients beside qdata.  A write to such a vector outside the operators drops the
    provenance -- an eager apply then reads the array itself, a replay would silently go on reading the old buffer.  CeedXGraphLaunch
    refuses instead; the buffers themselves are retired, never freed under a live graph."""
    ceed = cd.Ceed(product_lib, "/gpu/hip/mi355x")
    mesh = distorted_box(2, 2, 2)
    p = SolidProblem(ceed, mesh, 6, "hyperFS", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    n = p.lsize()
    X, Y, Yg = ceed.vector(n), ceed.vector(n), ceed.vector(n)
    u = p.smooth_state(0.1)
    X.set_array(u); p.form_residual(X, Y)
    X.set_array(np.random.default_rng(2).uniform(-1, 1, n)); X.device_pointer()
    p.apply_jacobian(p.fine, X, Y)
    assert "derived" in p.levels[p.fine].opJacob.kernel_name
    g = ceed.capture(lambda: p.apply_jacobian(p.fine, X, Yg))
    g.launch()
    assert np.array_equal(Yg.to_numpy(), Y.to_numpy())
    # the state overwritten by the caller: the eager apply falls back to the stored grad u, the replay is refused
    gu = p.gradu.to_numpy()
    p.gradu.set_array(gu)
    with pytest.raises(cd.CeedError, match="record the graph again"):
        g.launch()
    p.apply_jacobian(p.fine, X, Y)
    assert "derived" not in p.levels[p.fine].opJacob.kernel_name
    assert rel_err(Y.to_numpy(), Yg.to_numpy()) < 1e-12
    # the residual evaluated again: the derived state is valid again in the SAME buffer -> the old recording may replay
    Xu = ceed.vector(n).set_array(u)
    p.form_residual(Xu, Y)
    Yg.set_value(0.0)
    g.launch()
    p.apply_jacobian(p.fine, X, Y)
    assert np.array_equal(Yg.to_numpy(), Y.to_numpy())
    # qdata overwritten by the caller: the geometry is read from the array from now on, the recording is stale for good
    p.qdata.set_array(p.qdata.to_numpy())
    with pytest.raises(cd.CeedError, match="record the graph again"):
        g.launch()
    p.apply_jacobian(p.fine, X, Y)
    assert "qdata read" in p.levels[p.fine].opJacob.kernel_name
    assert rel_err(Y.to_numpy(), Yg.to_numpy()) < 1e-12
    g.destroy()
    p.destroy()


def test_halo_exchange_through_rccl_on_one_gpu(product_lib):
    """CeedXHalo* end to end on ONE GPU: a one-rank RCCL communicator whose only neighbour is the rank itself (ncclSend /
    ncclRecv to self inside one group is legal), so the library's pack kernel, RCCL's send and receive on the
    communicator's stream, the event hand-over between the two streams and the unpack-add kernel all run:
    y[idx] += y[idx].  (Two ranks cannot share this box's single GPU under RCCL; the multi-rank logic is covered by the
    gloo tests of the same neighbour lists.)"""
    import ctypes as C
    ceed = cd.Ceed(product_lib, "/gpu/hip/mi355x")
    L = ceed.L
    ident = C.create_string_buffer(128)
    L.chk(L.lib.CeedXCommGetUniqueId(ceed.h, ident))
    L.chk(L.lib.CeedXCommInit(ceed.h, 1, 0, ident))
    n = 200_003
    rng = np.random.default_rng(8)
    y0 = rng.uniform(-1, 1, n)
    lists = [np.sort(rng.choice(n, 50_000, replace=False)).astype(np.int32), rng.permutation(n)[:777].astype(np.int32), np.zeros(0, dtype=np.int32)]
    ranks = (C.c_int * 3)(0, 0, 0)
    counts = (C.c_int * 3)(*[a.size for a in lists])
    ptrs = (C.POINTER(C.c_int) * 3)(*[a.ctypes.data_as(C.POINTER(C.c_int)) for a in lists])
    h = C.c_void_p()
    # error returns of the constructor free what they had built: a neighbour outside the communicator,
    # a negative entry in the LAST list (everything before it already allocated) -- and a good call still works afterwards
    with pytest.raises(cd.CeedError):
        L.chk(L.lib.CeedXHaloCreate(ceed.h, 3, (C.c_int * 3)(0, 1, 0), counts, ptrs, C.byref(h)))
    bad = lists[1].copy(); bad[-1] = -5
    with pytest.raises(cd.CeedError):
        L.chk(L.lib.CeedXHaloCreate(ceed.h, 2, ranks, (C.c_int * 2)(lists[0].size, bad.size),
                                    (C.POINTER(C.c_int) * 2)(lists[0].ctypes.data_as(C.POINTER(C.c_int)), bad.ctypes.data_as(C.POINTER(C.c_int))), C.byref(h)))
    assert not h
    L.chk(L.lib.CeedXHaloCreate(ceed.h, 3, ranks, counts, ptrs, C.byref(h)))
    Y = ceed.vector(n).set_array(y0)
    want = y0.copy()
    for rep in range(3):
        L.chk(L.lib.CeedXHaloStart(h, Y.h))
        with pytest.raises(cd.CeedError):                     # one exchange in flight at a time
            L.chk(L.lib.CeedXHaloStart(h, Y.h))
        L.chk(L.lib.CeedXHaloFinish(h, Y.h))
        packed = [want[a].copy() for a in lists]              # all lists are packed BEFORE any is added
        for a, p in zip(lists, packed):
            want[a] += p
        assert np.array_equal(Y.to_numpy(), want), rep
    with pytest.raises(cd.CeedError):
        L.chk(L.lib.CeedXHaloFinish(h, Y.h))                  # nothing in flight
    # the same halo under the C++ harness: ApplyJacobian_Ceed ends with the interface sum (DMLocalToGlobal(ADD_VALUES), matops.c:57)
    from ceedpetscsolid_amd.harness import SolidApp
    mesh = distorted_box(3, 3, 2)
    app = SolidApp(ceed, mesh, 2, "hyperFS", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    nl = app.lsize()
    X, Y1, Y2 = ceed.vector(nl), ceed.vector(nl), ceed.vector(nl)
    X.set_array(np.linspace(0.0, 0.05, nl)); app.form_residual(X, Y1)
    X.set_array(rng.uniform(-1, 1, nl))
    app.apply_jacobian(app.fine, X, Y1)
    sub = np.sort(rng.choice(nl, nl // 3, replace=False)).astype(np.int32)
    hh = C.c_void_p()
    L.chk(L.lib.CeedXHaloCreate(ceed.h, 1, (C.c_int * 1)(0), (C.c_int * 1)(sub.size), (C.POINTER(C.c_int) * 1)(sub.ctypes.data_as(C.POINTER(C.c_int))), C.byref(hh)))
    app.set_halo(app.fine, hh)
    app.apply_jacobian(app.fine, X, Y2)
    y1, y2 = Y1.to_numpy(), Y2.to_numpy()
    expect = y1.copy(); expect[sub] += y1[sub]
    assert np.array_equal(y2, expect)
    app.set_halo(app.fine, None)
    app.apply_jacobian(app.fine, X, Y2)
    assert np.array_equal(Y2.to_numpy(), y1)
    L.chk(L.lib.CeedXHaloDestroy(C.byref(hh)))
    L.chk(L.lib.CeedXHaloDestroy(C.byref(h)))
    # bench.py's start-up check of the library's exchange (halo.checked_rccl_halo) on the one rank there is: no neighbours,
    # so both exchanges leave the test vector alone and the library's is accepted
    from ceedpetscsolid_amd.halo import HaloExchange, checked_rccl_halo
    from ceedpetscsolid_amd.mesh import build_dofmap
    dm = build_dofmap(mesh, 2)
    hx = HaloExchange(mesh, dm, device="cuda")
    got, note = checked_rccl_halo(ceed, hx, rng.uniform(-1, 1, dm.lsize), "cuda", timeout_s=60.0)
    assert got is not None and "checked" in note
    got.destroy()
    L.chk(L.lib.CeedXCommDestroy(ceed.h))


def test_unsupported_graphs_fail_loudly(gpu):
    with pytest.raises(cd.CeedError):
        gpu.qfunction("SomeUserQFunction", source="user.h:SomeUserQFunction")
    b = gpu.basis_lagrange(3, 3, 3, 3, cd.GAUSS)
    with pytest.raises(cd.CeedError):
        b.apply(1, cd.NOTRANSPOSE, cd.EVAL_INTERP, gpu.vector(81), gpu.vector(81))


def test_split_phase_apply_equals_full_apply(gpu):
    """CeedXOperatorApplyPhase 0 then 1 == CeedOperatorApply (communication-overlap form, used at N > 1)."""
    from ceedpetscsolid_amd.mesh import build_dofmap, reorder_elements_first
    mesh = distorted_box(4, 4, 4)
    zc = mesh.coords[mesh.cells][:, :, 2]
    lead = (zc.min(axis=1) < 0.5 + 1e-9) & (zc.max(axis=1) > 0.5 - 1e-9)       # elements touching the plane z = 0.5
    lead_far = np.abs(mesh.coords[mesh.cells][:, :, 2] - 0.5).min(axis=1) < 0.2  # (distortion tolerance)
    mesh = reorder_elements_first(mesh, lead_far)
    p = SolidProblem(gpu, mesh, 3, "hyperFS", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    lv = p.levels[p.fine]
    n = p.lsize()
    # priority nodes: every node all of whose elements are leading elements and that lies near the plane
    nlead = int(lead_far.sum())
    touched_by_rest = np.zeros(lv.dofmap.nnodes, dtype=bool)
    touched_by_rest[lv.dofmap.elem_nodes[nlead:].ravel()] = True
    near = np.abs(lv.dofmap.node_coords[:, 2] - 0.5) < 0.05
    prio_nodes = near & ~touched_by_rest
    assert prio_nodes.sum() > 0
    prio = np.repeat(prio_nodes.astype(np.uint8), 3)
    X, Y, Y2 = gpu.vector(n), gpu.vector(n), gpu.vector(n)
    X.set_array(p.smooth_state(0.1)); p.form_residual(X, Y)
    X.set_array(np.random.default_rng(2).uniform(-1, 1, n))
    p.apply_jacobian(p.fine, X, Y)
    op = lv.opJacob
    op.set_overlap_split(nlead, prio)
    Y2.set_value(7.0)
    op.apply_phase(X, Y2, 0)
    y0 = Y2.to_numpy()
    assert np.array_equal(y0[prio == 1], Y.to_numpy()[prio == 1])               # interface nodes complete after phase 0
    op.apply_phase(X, Y2, 1)
    assert np.array_equal(Y2.to_numpy(), Y.to_numpy())                          # deterministic scatter: bitwise
    with pytest.raises(cd.CeedError):                                            # contract check
        op.set_overlap_split(1, prio)


def test_apply_is_bitwise_reproducible(gpu):
    """The E-vector + k_assemble scatter sums each node's contributors in a fixed (element) order:
    repeated applies are bitwise identical (f64 atomics would not be)."""
    mesh = hollow_cylinder_mesh(3, 12, 6)
    p = SolidProblem(gpu, mesh, 4, "hyperFS", nu=0.3, E=1.0, bc_sides=[998, 999], multigrid="none")
    n = p.lsize()
    X, Y = gpu.vector(n), gpu.vector(n)
    X.set_array(p.smooth_state(0.1)); p.form_residual(X, Y)
    r0 = Y.to_numpy()
    X.set_array(np.random.default_rng(4).uniform(-1, 1, n))
    outs = []
    for _ in range(4):
        p.apply_jacobian(p.fine, X, Y)
        outs.append(Y.to_numpy())
    assert all(np.array_equal(o, outs[0]) for o in outs[1:])
    X.set_array(p.smooth_state(0.1)); p.form_residual(X, Y)
    assert np.array_equal(Y.to_numpy(), r0)


@pytest.mark.parametrize("mk,degree,problem", [
    (lambda: hollow_cylinder_mesh(4, 24, 16), 4, "hyperFS"),    # 1 536 elements: 96 groups per XCD chunk, three buckets each
    (lambda: hollow_cylinder_mesh(10, 40, 20), 4, "hyperFS"),   # 8 000 elements
    (lambda: distorted_box(12, 12, 12), 2, "hyperSS"),          # Q = 3: four elements per group
    (lambda: distorted_box(9, 7, 5), 1, "linElas"),             # Q = 2: eight elements per group, ragged last group
    (lambda: distorted_box(7, 6, 6), 6, "hyperFS"),             # Q = 7: one element per group
], ids=["cyl1536 p4", "cyl8000 p4", "box p2", "box p1", "box p6"])
@pytest.mark.parametrize("mode", ["pipelined"])
def test_pipelined_assembly_equals_serial_assembly_bitwise(product_lib, mk, degree, problem, mode):
    """The pipelined form of the restriction transpose (default on large launches): the apply cut into segments, the rows of
    segment k summed by a k_assemble launch beside the fused kernel of segment k + 1 -- against the serial form (k_assemble
    after the fused kernel, CEED_MI355X_ASSEMBLE=serial): same E-vector values, same element order, so the results are
    BITWISE equal.  The inputs alternate between applies, so an E-vector entry read before its producer had finished would
    show as the previous apply's value.  (Round 2's dependent in-kernel forms -- gated, folded, dynamic -- were measured
    -3 ... +5 % and left the tree in round 3.)"""
    mesh = mk()
    # three segments whatever the mesh size (the default asks for four rounds of the waves per segment)
    os.environ["CEED_MI355X_PIPE_MIN_ROUNDS"], os.environ["CEED_MI355X_PIPE_SEGMENTS"] = "0", "3"
    try:
        gated = _ceed_with_env(product_lib, "CEED_MI355X_ASSEMBLE", mode)
    finally:
        os.environ.pop("CEED_MI355X_PIPE_MIN_ROUNDS", None); os.environ.pop("CEED_MI355X_PIPE_SEGMENTS", None)
    serial = _ceed_with_env(product_lib, "CEED_MI355X_ASSEMBLE", "serial")
    probs = [SolidProblem(c, mesh, degree, problem, nu=0.3, E=1.0, bc_sides=[sorted(mesh.side_sets)[0]], multigrid="none") for c in (gated, serial)]
    n = probs[0].lsize()
    rng = np.random.default_rng(11)
    vecs = [(c.vector(n), c.vector(n)) for c in (gated, serial)]
    u0 = probs[0].smooth_state(0.1)
    for (X, Y), p in zip(vecs, probs):
        X.set_array(u0); p.form_residual(X, Y)
    assert np.array_equal(vecs[0][1].to_numpy(), vecs[1][1].to_numpy())
    for it in range(12):
        x = rng.uniform(-1, 1, n) * (10.0 ** rng.integers(-3, 4))
        outs = []
        for (X, Y), p in zip(vecs, probs):
            X.set_array(x)
            p.apply_jacobian(p.fine, X, Y)
            if it % 3 == 2:   # back-to-back applies without a host round trip in between
                p.apply_jacobian(p.fine, X, Y); p.apply_jacobian(p.fine, X, Y)
            outs.append(Y.to_numpy())
        assert np.array_equal(outs[0], outs[1]), f"apply {it}"
    assert probs[0].levels[probs[0].fine].opJacob.kernel_name == probs[1].levels[probs[1].fine].opJacob.kernel_name
    info = [p.levels[p.fine].opJacob.launch_info() for p in probs]
    assert info[1]["segments"] == 1
    if mode == "pipelined":
        assert info[0]["segments"] == 3 and info[0]["streams"] == 2 and info[0]["assemble_launches"] == 3
        # recorded into a graph (fork to the second stream and join inside the capture) and replayed on new data
        (X, Y), p = vecs[0], probs[0]
        Yg = gated.vector(n)
        g = gated.capture(lambda: p.apply_jacobian(p.fine, X, Yg))
        for _ in range(3):
            x = rng.uniform(-1, 1, n)
            for (Xs, Ys), ps in zip(vecs, probs):
                Xs.set_array(x); Xs.device_pointer()
            probs[1].apply_jacobian(probs[1].fine, vecs[1][0], vecs[1][1])
            Yg.set_value(-7.0)
            g.launch()
            assert np.array_equal(Yg.to_numpy(), vecs[1][1].to_numpy())
        g.destroy()


def _tangent_properties(probs, vecs, levels, rng, n_of, bitwise_label):
    """Size-independent properties of the Jacobian apply on every level in `levels`, for the problems in `probs` (the first is the
    one under test, the others must agree with it BITWISE): symmetry of the tangent, linearity, rigid translations in the null
    space (no BCs)."""
    for lv in levels:
        n = n_of(lv)
        v, w = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)

        def J(z):
            outs = []
            for (X, Y), p in zip(vecs[lv], probs):
                X.set_array(z); p.apply_jacobian(lv, X, Y); outs.append(Y.to_numpy())
            for o in outs[1:]:
                assert np.array_equal(o, outs[0]), f"{bitwise_label}, level {lv}"
            return outs[0]
        jv, jw = J(v), J(w)
        assert abs(v @ jw - w @ jv) < 1e-11 * abs(v @ jw), lv
        assert rel_err(J(2.0 * v - 0.5 * w), 2.0 * jv - 0.5 * jw) < 1e-12, lv
        t = np.tile([0.3, -1.0, 2.0], n // 3)
        assert np.abs(J(t)).max() < 1e-11 * np.abs(jv).max(), lv


def test_full_size_properties_config5_whole_box(product_lib):
    """The WHOLE of BASELINE config 5 on one GPU -- box 64^3, degree 6, hyperFS: 262 144 hexes, 170 M dofs, the apply pipelined in
    nine segments over two streams (15 in round 3; VERDICT r3 item 4: benchmarked then, never checked).  Far beyond the oracle, so: the
    pipelined apply BITWISE equal to the serial form (k_assemble after one fused launch) on the same inputs, and the
    size-independent properties of the tangent (symmetry, linearity, rigid-translation null space)."""
    mesh = box_mesh(64, 64, 64)
    pipe = cd.Ceed(product_lib, "/gpu/hip/mi355x")
    serial = _ceed_with_env(product_lib, "CEED_MI355X_ASSEMBLE", "serial")
    probs = [SolidProblem(c, mesh, 6, "hyperFS", nu=0.3, E=1.0, multigrid="none") for c in (pipe, serial)]
    n = probs[0].lsize()
    assert n == 3 * 385 ** 3
    vecs = {0: [(c.vector(n), c.vector(n)) for c in (pipe, serial)]}
    u = probs[0].smooth_state(0.1)
    res = []
    for (X, Y), p in zip(vecs[0], probs):
        X.set_array(u); p.form_residual(X, Y); res.append(Y.to_numpy())
    assert np.array_equal(res[0], res[1])
    del res
    _tangent_properties(probs, vecs, [0], np.random.default_rng(17), lambda lv: n, "pipelined (9 segments) vs serial")
    info = [p.levels[0].opJacob.launch_info() for p in probs]
    assert info[0]["segments"] >= 8 and info[0]["streams"] == 2 and info[1]["segments"] == 1, info
    for p in probs:
        p.destroy()


def test_properties_unstructured_cylinder_all_levels_pipelined(product_lib):
    """The reference's cylinder8_44928e_2ss_us (CUBIT element order) at degree 4 with its multigrid ladder (levels of degree
    1, 2, 4, all on the fine quadrature): every level's Jacobian apply forced into the pipelined form (44 928 hexes are
    10.97 rounds of the waves: two segments with a non-trivial remainder in the first) against the serial form, BITWISE, plus the
    tangent's properties on every level."""
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_44928e_2ss_us.npz"))
    os.environ["CEED_MI355X_PIPE_MIN_TOTAL"] = "0"
    try:
        pipe = cd.Ceed(product_lib, "/gpu/hip/mi355x")
    finally:
        os.environ.pop("CEED_MI355X_PIPE_MIN_TOTAL", None)
    serial = _ceed_with_env(product_lib, "CEED_MI355X_ASSEMBLE", "serial")
    probs = [SolidProblem(c, mesh, 4, "hyperFS", nu=0.3, E=1.0) for c in (pipe, serial)]
    nl = len(probs[0].levels)
    assert [lv.degree for lv in probs[0].levels] == [1, 2, 4]
    vecs = {lv: [(c.vector(probs[0].lsize(lv)), c.vector(probs[0].lsize(lv))) for c in (pipe, serial)] for lv in range(nl)}
    u = probs[0].smooth_state(0.1)
    for (X, Y), p in zip(vecs[nl - 1], probs):
        X.set_array(u); p.form_residual(X, Y)
    _tangent_properties(probs, vecs, range(nl), np.random.default_rng(23), lambda lv: probs[0].lsize(lv), "pipelined vs serial")
    for lv in range(nl):
        info = [p.levels[lv].opJacob.launch_info() for p in probs]
        assert info[0]["segments"] >= 2 and info[1]["segments"] == 1, (lv, info)
    for p in probs:
        p.destroy()


def test_recorded_graph_survives_growth_of_the_scratch(product_lib):
    """A recorded graph has the E-vector scratch pointer of its capture time in its kernel nodes.  Applying a LARGER
    operator on the same Ceed afterwards (a second problem, a finer level) grows the scratch: the old buffer must stay
    alive for the graph's replays (ceed_need_evec parks it), and growth DURING a capture must not break the capture."""
    ceed = cd.Ceed(product_lib, "/gpu/hip/mi355x")      # a Ceed of its own: its scratch starts empty
    small = SolidProblem(ceed, distorted_box(2, 2, 2), 2, "hyperFS", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    n = small.lsize()
    X, Y, Yg = ceed.vector(n), ceed.vector(n), ceed.vector(n)
    X.set_array(small.smooth_state(0.1)); small.form_residual(X, Y)
    X.set_array(np.random.default_rng(5).uniform(-1, 1, n))
    small.apply_jacobian(small.fine, X, Y)
    eager = Y.to_numpy()
    g = ceed.capture(lambda: small.apply_jacobian(small.fine, X, Yg))
    big = SolidProblem(ceed, distorted_box(4, 4, 3), 4, "hyperFS", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    nb = big.lsize()
    Xb, Yb, Yb2 = ceed.vector(nb), ceed.vector(nb), ceed.vector(nb)
    Xb.set_array(big.smooth_state(0.1)); big.form_residual(Xb, Yb)      # first apply of the larger restriction: the scratch grows
    Xb.set_array(np.random.default_rng(6).uniform(-1, 1, nb))
    big.apply_jacobian(big.fine, Xb, Yb)
    for _ in range(2):
        Yg.set_value(-3.0)
        g.launch()
        assert np.array_equal(Yg.to_numpy(), eager)
    # a COLD operator cannot be recorded (its transpose map is built on the host at the first apply): a clear error, and
    # the Ceed, the older graph and the parked scratch stay usable
    cold = SolidProblem(ceed, distorted_box(5, 4, 4), 4, "hyperFS", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    nh = cold.lsize()
    Xh, Yh = ceed.vector(nh), ceed.vector(nh)
    Xh.set_array(cold.smooth_state(0.1)); Xh.device_pointer(); Yh.set_value(0.0)
    with pytest.raises(cd.CeedError, match="once before recording"):
        ceed.capture(lambda: cold.form_residual(Xh, Yh))
    cold.form_residual(Xh, Yh)                                          # eagerly: the scratch grows again
    g2 = ceed.capture(lambda: big.apply_jacobian(big.fine, Xb, Yb2))    # a second, larger recording beside the first
    Yb2.set_value(-1.0); Yg.set_value(-1.0)
    g2.launch(); g.launch()
    assert np.array_equal(Yb2.to_numpy(), Yb.to_numpy())
    assert np.array_equal(Yg.to_numpy(), eager)
    g.destroy(); g2.destroy()


def test_apply_add_and_empty_vectors(gpu):
    """CeedOperatorApplyAdd accumulates (y += J x); zero-length vectors are legal objects."""
    import ctypes as C
    mesh = distorted_box(2, 2, 1)
    p = SolidProblem(gpu, mesh, 2, "linElas", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    n = p.lsize()
    x = np.random.default_rng(9).uniform(-1, 1, n)
    X, Y = gpu.vector(n).set_array(x), gpu.vector(n)
    p.apply_jacobian(p.fine, X, Y)
    jx = Y.to_numpy()
    y0 = np.random.default_rng(10).uniform(-1, 1, n)
    Y.set_array(y0)
    L = gpu.L
    op = p.levels[p.fine].opJacob
    L.chk(L.lib.CeedOperatorApplyAdd(op.h, X.h, Y.h, C.c_void_p(L.REQUEST_IMMEDIATE)))
    assert rel_err(Y.to_numpy(), y0 + jx) < 1e-14
    e = gpu.vector(0)
    assert e.to_numpy().size == 0


def _ceed_with_env(product_lib, key, val):
    old = os.environ.get(key)
    os.environ[key] = val
    try:
        return cd.Ceed(product_lib, "/gpu/hip/mi355x")      # the switches are read at CeedInit
    finally:
        if old is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = old


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["linElas", "hyperSS", "hyperFS"])
def test_recomputed_geometry_equals_stored_qdata(gpu, product_lib, problem):
    """The pencil kernel recomputes SetupGeo's factors (common.h:47-101) per point from the element's trilinear map
    when the qdata vector still is what the SetupGeo operator wrote (FusedGradArgs::geo); CEED_MI355X_GEO=0 reads the
    stored 10 values per point instead.  Same numbers to rounding on distorted (non-affine) elements, every level."""
    plain = _ceed_with_env(product_lib, "CEED_MI355X_GEO", "0")
    for mesh, degree in ((distorted_box(3, 2, 3, seed=2, amp=0.2), 4), (distorted_box(2, 2, 1, seed=3, amp=0.2), 6),
                         (distorted_box(5, 3, 1, seed=5, amp=0.2), 1), (hollow_cylinder_mesh(2, 8, 3), 3)):
        outs = []
        for c in (gpu, plain):
            p = SolidProblem(c, mesh, degree, problem, nu=0.3, E=2.0, bc_sides=[1] if 1 in mesh.side_sets else [998])
            n = p.lsize()
            X, R = c.vector(n), c.vector(n)
            X.set_array(p.smooth_state(0.1)); p.form_residual(X, R)
            res = [R.to_numpy()]
            for lv in range(len(p.levels)):
                nl = p.lsize(lv)
                x = c.vector(nl).set_array(np.random.default_rng(7 + lv).uniform(-1, 1, nl))
                y = c.vector(nl)
                p.apply_jacobian(lv, x, y)
                res.append(y.to_numpy())
            outs.append(res)
        for a, b in zip(*outs):
            assert rel_err(a, b) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["linElas", "hyperSS", "hyperFS"])
def test_affine_elements_take_the_per_element_factors(gpu, oracle, product_lib, problem):
    """On a mesh whose elements are ALL affine (parallelepipeds: the cube of config 2, the boxes of configs 1 and 5) dXdx and
    det J are constants of an element: the SetupGeo operator finds that out (12 vanishing map coefficients per element), and
    the fused kernels then read ten numbers per element instead of forming J, its adjugate and a reciprocal at every point
    (qfunctions/common.h:47-101 is the arithmetic either way).  Same results as the general per-point recompute
    (CEED_MI355X_AFFINE=0) and as the oracle on sheared, stretched boxes, every level; a mesh with ONE non-affine element
    takes the general path everywhere."""
    general = _ceed_with_env(product_lib, "CEED_MI355X_AFFINE", "0")
    shear = np.array([[1.0, 0.3, -0.2], [0.1, 0.7, 0.25], [-0.15, 0.2, 1.4]])
    def sheared(nx, ny, nz):
        m = box_mesh(nx, ny, nz)
        m.coords = m.coords @ shear.T + np.array([0.3, -0.1, 0.2])
        return m
    mixed = sheared(3, 3, 2)
    mixed.coords = mixed.coords.copy()
    mixed.coords[0] += np.array([0.05, -0.03, 0.02])         # one corner moved: its element is no parallelepiped any more
    for mesh, degree, want in ((sheared(3, 2, 3), 4, "affine elements"), (sheared(2, 2, 1), 6, "affine elements"),
                               (sheared(5, 3, 2), 2, "affine elements"), (sheared(4, 3, 3), 1, "affine elements"),
                               (mixed, 3, "recomputed per point")):
        outs = []
        for c in (gpu, general, oracle):
            p = SolidProblem(c, mesh, degree, problem, nu=0.3, E=2.0, bc_sides=[1])
            n = p.lsize()
            X, R = c.vector(n), c.vector(n)
            X.set_array(p.smooth_state(0.1)); p.form_residual(X, R)
            res = [R.to_numpy()]
            for lv in range(len(p.levels)):
                nl = p.lsize(lv)
                x = c.vector(nl).set_array(np.random.default_rng(7 + lv).uniform(-1, 1, nl))
                y = c.vector(nl)
                p.apply_jacobian(lv, x, y)
                res.append(y.to_numpy())
            outs.append(res)
            if c is gpu:
                assert want in p.levels[p.fine].opJacob.kernel_name, p.levels[p.fine].opJacob.kernel_name
            if c is general:
                assert "recomputed per point" in p.levels[p.fine].opJacob.kernel_name
        for a, b, o in zip(*outs):
            assert rel_err(a, b) < 1e-13
            assert rel_err(a, o) < 1e-10


def _relabel_axes(mesh, perm):
    """The same mesh with every element's local (reference) directions relabelled: new direction d is old direction perm[d]
    (a cyclic perm keeps the Jacobians positive); vertex order and local face numbers of the side sets follow."""
    import copy
    m = copy.copy(mesh)
    old_of_new = [sum(((c >> d) & 1) << perm[d] for d in range(3)) for c in range(8)]
    m.cells = mesh.cells[:, old_of_new].copy()
    new_dir = {perm[d]: d for d in range(3)}
    m.side_sets = {k: np.stack([v[:, 0], 2 * np.array([new_dir[f // 2] for f in v[:, 1]]) + v[:, 1] % 2], axis=1) for k, v in mesh.side_sets.items()}
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["linElas", "hyperSS", "hyperFS"])
def test_swept_elements_take_the_two_by_two_jacobian(gpu, oracle, product_lib, problem):
    """The reference's cylinders are EXTRUDED meshes: in every element x and y are bilinear in two reference directions and z is
    linear in the third (tests/golden/mesh_cylinder8_*: the sweep runs along the element's zeta in two of them, along eta in
    cylinder8_44928e).  The SetupGeo operator finds that out (one sweep direction for the whole mesh), and the fused kernels then
    form a 2 x 2 Jacobian per point and multiply with the five entries of dXdx that are left (qfunctions/common.h:47-101 is the
    arithmetic either way).  Same results as the general per-point recompute (CEED_MI355X_SWEPT=0) and as the oracle for the
    sweep along each of the three reference directions, every level; a mesh with ONE element that is no prism takes the general
    path everywhere; the reference's own unstructured cylinder is recognised."""
    general = _ceed_with_env(product_lib, "CEED_MI355X_SWEPT", "0")
    base = hollow_cylinder_mesh(2, 8, 3)
    mixed = hollow_cylinder_mesh(2, 6, 2)
    mixed.coords = mixed.coords.copy()
    mixed.coords[0] += np.array([0.0, 0.0, 0.07])              # one vertex lifted: its elements are no prisms any more
    real = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    cases = [(base, 4, "swept elements"), (_relabel_axes(base, [1, 2, 0]), 4, "swept elements"), (_relabel_axes(base, [2, 0, 1]), 3, "swept elements"),
             (_relabel_axes(hollow_cylinder_mesh(2, 6, 2), [1, 2, 0]), 6, "swept elements"), (real, 2, "swept elements"), (mixed, 3, "dXdx recomputed per point")]
    for mesh, degree, want in cases:
        outs = []
        for c in (gpu, general, oracle):
            p = SolidProblem(c, mesh, degree, problem, nu=0.3, E=2.0, bc_sides=[998])
            n = p.lsize()
            X, R = c.vector(n), c.vector(n)
            X.set_array(p.smooth_state(0.1)); p.form_residual(X, R)
            res = [R.to_numpy()]
            for lv in range(len(p.levels)):
                nl = p.lsize(lv)
                x = c.vector(nl).set_array(np.random.default_rng(7 + lv).uniform(-1, 1, nl))
                y = c.vector(nl)
                p.apply_jacobian(lv, x, y)
                res.append(y.to_numpy())
            outs.append(res)
            name = p.levels[p.fine].opJacob.kernel_name if c is not oracle else ""
            if c is gpu:
                assert want in name and ("swept" in name) == (want == "swept elements"), name
            if c is general:
                assert "recomputed per point" in name and "swept" not in name, name
        for a, b, o in zip(*outs):
            assert rel_err(a, b) < 1e-13
            assert rel_err(a, o) < 1e-10


@pytest.mark.gpu
def test_derived_state_of_the_finite_strain_tangent(gpu, oracle, product_lib):
    """HyperFSF leaves, beside the stored grad u, the DERIVED state HyperFSdF needs -- F^-1 and lambda ln J - mu, ten doubles
    per point -- and the Jacobian kernels read that instead of forming an adjugate, a determinant, a reciprocal and the log
    series at every point of every apply (qfunctions_device.hpp; hyperFS.h:286-464 is the map).  Same numbers as the plain
    form (CEED_MI355X_DERIVED=0) and as the oracle, every level, nu up to 0.49; and the derived state is DROPPED when the
    application writes grad u itself: the Jacobian must then follow the new values."""
    plain = _ceed_with_env(product_lib, "CEED_MI355X_DERIVED", "0")
    for mesh, degree, nu, amp in ((distorted_box(3, 2, 3, seed=2, amp=0.2), 5, 0.3, 0.1), (distorted_box(2, 2, 1, seed=3, amp=0.2), 6, 0.49, 0.05),
                                  (hollow_cylinder_mesh(2, 6, 2), 7, 0.3, 0.2), (hollow_cylinder_mesh(2, 8, 3), 4, 0.3, 0.3)):
        outs, probs = [], []
        for c in (gpu, plain, oracle):
            p = SolidProblem(c, mesh, degree, "hyperFS", nu=nu, E=2.0, bc_sides=[1] if 1 in mesh.side_sets else [998])
            n = p.lsize()
            X, R = c.vector(n), c.vector(n)
            X.set_array(p.smooth_state(amp)); p.form_residual(X, R)
            res = [R.to_numpy()]
            for lv in range(len(p.levels)):
                nl = p.lsize(lv)
                x = c.vector(nl).set_array(np.random.default_rng(7 + lv).uniform(-1, 1, nl))
                y = c.vector(nl)
                p.apply_jacobian(lv, x, y)
                res.append(y.to_numpy())
            outs.append(res); probs.append(p)
        # used from Q = 6 on (where it measured a gain: kernels.hpp, pencil_derived_state); below, the plain tangent
        assert ("HyperFSdF+derived" in probs[0].levels[probs[0].fine].opJacob.kernel_name) == (degree + 1 >= 6)
        assert "derived" not in probs[1].levels[probs[1].fine].opJacob.kernel_name
        for a, b, o in zip(*outs):
            assert rel_err(a, b) < 1e-13
            assert rel_err(a, o) < 1e-10
        # the application overwrites grad u (half of it): both libraries must now linearise about THAT state
        ys = []
        for c, p in ((gpu, probs[0]), (plain, probs[1])):
            p.gradu.set_array(0.5 * p.gradu.to_numpy())
            nl = p.lsize()
            x = c.vector(nl).set_array(np.random.default_rng(99).uniform(-1, 1, nl))
            y = c.vector(nl)
            p.apply_jacobian(p.fine, x, y)
            ys.append(y.to_numpy())
        assert "derived" not in probs[0].levels[probs[0].fine].opJacob.kernel_name
        assert rel_err(ys[0], ys[1]) < 1e-13 and rel_err(ys[0], outs[0][-1]) > 1e-3


@pytest.mark.gpu
def test_overwritten_qdata_is_read_not_recomputed(gpu):
    """The recompute is only valid while qdata is SetupGeo's output: any other write to the vector must switch the
    operators back to reading it.  Doubling all ten entries multiplies the linear-elastic action by 2 (w detJ) x 2 x 2
    (dXdx on both sides)."""
    mesh = distorted_box(2, 2, 2, seed=1, amp=0.1)
    p = SolidProblem(gpu, mesh, 3, "linElas", nu=0.3, E=1.0, bc_sides=[1], multigrid="none")
    n = p.lsize()
    x = gpu.vector(n).set_array(np.random.default_rng(0).uniform(-1, 1, n))
    y1, y2 = gpu.vector(n), gpu.vector(n)
    p.apply_jacobian(p.fine, x, y1)
    p.qdata.set_array(2.0 * p.qdata.to_numpy())
    p.apply_jacobian(p.fine, x, y2)
    assert rel_err(y2.to_numpy(), 8.0 * y1.to_numpy()) < 1e-13


@pytest.mark.gpu
def test_direct_interior_stores_equal_the_assembled_path(gpu, product_lib):
    """Element-interior nodes have one contributor: the pencil kernel stores them straight into y and keeps a shell-only
    E-vector (FusedGradArgs::direct).  CEED_MI355X_DIRECT=0 sends every node through the E-vector; the two must give
    the SAME numbers (a single-term sum is exact), with Dirichlet rows, on every level, split-phase included."""
    old = os.environ.get("CEED_MI355X_DIRECT")
    os.environ["CEED_MI355X_DIRECT"] = "0"
    try:
        plain = cd.Ceed(product_lib, "/gpu/hip/mi355x")      # read at CeedInit
    finally:
        if old is None:
            os.environ.pop("CEED_MI355X_DIRECT", None)
        else:
            os.environ["CEED_MI355X_DIRECT"] = old
    for mesh, degree in ((distorted_box(3, 2, 3, seed=2), 4), (distorted_box(2, 2, 1, seed=3), 6), (distorted_box(5, 1, 1), 2),
                         (hollow_cylinder_mesh(3, 8, 5), 4), (distorted_box(7, 3, 1, seed=5), 3)):
        outs = []
        for c in (gpu, plain):
            p = SolidProblem(c, mesh, degree, "hyperFS", nu=0.3, E=2.0, bc_sides=[sorted(mesh.side_sets)[0]])
            n = p.lsize()
            X, R = c.vector(n), c.vector(n)
            X.set_array(p.smooth_state(0.1)); p.form_residual(X, R)
            res = [R.to_numpy()]
            for lv in range(len(p.levels)):
                nl = p.lsize(lv)
                x = c.vector(nl).set_array(np.random.default_rng(7 + lv).uniform(-1, 1, nl))
                y = c.vector(nl).set_value(3.0)               # overwritten, interior nodes included
                p.apply_jacobian(lv, x, y)
                res.append(y.to_numpy())
            outs.append(res)
        for a, b in zip(*outs):
            assert np.array_equal(a, b)


def _forcing_and_true(c, p, kind):
    """opSetupForce / opTrue as the reference wires them (setuplibceed.c:555-583, 608-636)."""
    lv = p.levels[p.fine]
    n = p.lsize()
    if kind == "true":
        qf = c.qfunction("MMSTrueSoln", source="qfunctions/manufacturedTrue.h:MMSTrueSoln")
        qf.add_input("x", 3, cd.EVAL_INTERP).add_output("true_soln", 3, cd.EVAL_NONE)
        bxt = c.basis_lagrange(3, 3, 2, lv.degree + 1, cd.GAUSS_LOBATTO)      # basisxtrue, :600-603
        op = c.operator(qf)
        op.set_field("x", p.Erestrictx, bxt, "active")
        op.set_field("true_soln", lv.Erestrictu, None, "active")
    else:
        name = "SetupMMSForce" if kind == "mms" else "SetupConstantForce"
        src = "manufacturedForce.h" if kind == "mms" else "constantForce.h"
        qf = c.qfunction(name, source=f"qfunctions/{src}:{name}")
        qf.add_input("x", 3, cd.EVAL_INTERP).add_input("qdata", 10, cd.EVAL_NONE).add_output("force", 3, cd.EVAL_INTERP)
        if kind == "mms":
            qf.set_context(p.phys)
        else:
            _forcing_and_true.vec = np.array([0.3, -1.0, 2.0])
            qf.set_context(_forcing_and_true.vec, reported_size=8)             # sizeof(*forcingVector) quirk, :565-566
        op = c.operator(qf)
        op.set_field("x", p.Erestrictx, p.basisx, "active")
        op.set_field("qdata", p.Erestrictqdi, None, p.qdata)
        op.set_field("force", lv.Erestrictu, lv.basisu, "active")
    F = c.vector(n)
    op.apply(p.xcoord, F)
    return F.to_numpy()



@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["const", "mms", "true"])
def test_forcing_and_true_solution_operators_match_oracle(oracle, gpu, kind):
    mesh = hollow_cylinder_mesh(2, 8, 3)
    fo, fg = (_forcing_and_true(c, SolidProblem(c, mesh, 3, "linElas", nu=0.3, E=1e6, bc_sides=[998], multigrid="none"), kind)
              for c in (oracle, gpu))
    assert np.linalg.norm(fo) > 0 and rel_err(fg, fo) < TOL


@pytest.mark.gpu
def test_config1_mms_on_the_device(gpu):
    """BASELINE config 1 end to end on the MI355X path: linElas, box 4x4x4, degree 2, SetupMMSForce forcing,
    BCMMS boundary values, one Newton step of the PCG-pMG solver; the reference's own acceptance gate is a
    relative L2 error <= 0.05 against MMSTrueSoln (elasticity.c:790-810)."""
    from ceedpetscsolid_amd.solver import NewtonPMG
    mesh = box_mesh(4, 4, 4)
    p = SolidProblem(gpu, mesh, 2, "linElas", nu=0.3, E=1e6, bc_all_boundary=True)
    lv = p.levels[p.fine]
    f = _forcing_and_true(gpu, p, "mms")
    ut = _forcing_and_true(gpu, p, "true")
    mult = gpu.vector(p.lsize()); lv.Erestrictu.multiplicity(mult)
    ut = ut / mult.to_numpy()                                                   # setuplibceed.c:626-636
    s = NewtonPMG(p, mms=True, forcing=f)
    st = s.solve(1)
    assert st.converged and st.newton_its == 1
    u = s.U.to_numpy() + s.bc_values(1.0)
    err = np.linalg.norm(u - ut) / np.linalg.norm(ut)
    assert err < 0.05 and err < 5e-3, err


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["linElas", "hyperSS", "hyperFS"])
def test_strain_energy_operator_matches_oracle(oracle, gpu, problem):
    """opEnergy / ComputeStrainEnergy (setuplibceed.c:651-670, matops.c:247-296) on the device."""
    from ceedpetscsolid_amd.postprocess import StrainEnergy
    mesh = hollow_cylinder_mesh(2, 8, 3)
    res = []
    for c in (oracle, gpu):
        p = SolidProblem(c, mesh, 3, problem, nu=0.3, E=1e3, bc_sides=[998], multigrid="none")
        se = StrainEnergy(p, problem)
        X = c.vector(p.lsize()).set_array(p.smooth_state(0.05))
        res.append((se.compute(X), se.eloc.to_numpy()))
    assert abs(res[1][0] - res[0][0]) < TOL * abs(res[0][0]) and rel_err(res[1][1], res[0][1]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["linElas", "hyperSS", "hyperFS"])
def test_diagnostic_operator_matches_oracle(oracle, gpu, problem):
    """opDiagnostic (setuplibceed.c:679-737) incl. its SetupGeo on the GLL points, on the device."""
    from ceedpetscsolid_amd.postprocess import Diagnostics
    mesh = hollow_cylinder_mesh(2, 8, 3)
    res = []
    for c in (oracle, gpu):
        p = SolidProblem(c, mesh, 3, problem, nu=0.3, E=1e3, bc_sides=[998], multigrid="none")
        d = Diagnostics(p, problem)
        res.append(d.compute(c.vector(p.lsize()).set_array(p.smooth_state(0.05))))
    for k in range(8):
        assert rel_err(res[1][:, k], res[0][:, k]) < TOL, k


@pytest.mark.gpu
def test_reference_testargs_case_on_the_device(gpu):
    """The reference's ONLY scripted test, elasticity.c:36: `-test -degree 3 -nu 0.3 -E 1 -dm_plex_box_faces 3,3,3`
    (MMS forcing, BCMMS on the whole boundary); it passes when the relative L2 error is <= 0.05 (:807-811)."""
    from ceedpetscsolid_amd.solver import NewtonPMG
    p = SolidProblem(gpu, box_mesh(3, 3, 3), 3, "linElas", nu=0.3, E=1.0, bc_all_boundary=True)
    lv = p.levels[p.fine]
    f = _forcing_and_true(gpu, p, "mms")
    ut = _forcing_and_true(gpu, p, "true")
    mult = gpu.vector(p.lsize()); lv.Erestrictu.multiplicity(mult)
    ut = ut / mult.to_numpy()
    s = NewtonPMG(p, mms=True, forcing=f)
    st = s.solve(1)
    assert st.converged
    err = np.linalg.norm(s.U.to_numpy() + s.bc_values(1.0) - ut) / np.linalg.norm(ut)
    assert err <= 0.05, err
