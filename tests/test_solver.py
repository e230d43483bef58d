"""Newton - CG - p-multigrid plumbing: convergence on the oracle (CPU) and agreement
of the device solve with the oracle solve (GPU)."""
import os

import numpy as np
import pytest

from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, load_mesh_npz
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG, bc_clamp
from conftest import GOLDEN, rel_err

CLAMP = {998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}


def test_bc_clamp_translation_and_rotation():
    X = np.array([[1.0, 2.0, 3.0], [0.5, -1.0, 0.25]])
    assert np.allclose(bc_clamp(X, 0.5, translate=(2, 4, 6)), [[1, 2, 3], [1, 2, 3]])
    # boundary.c:65-71: angle in units of pi; y/z components are the Rodrigues rotation about z
    u = bc_clamp(X, 1.0, axis=(0, 0, 1), angle_over_pi=0.5)
    assert np.allclose(u[:, 1], X[:, 0] - X[:, 1]) and np.allclose(u[:, 2], 0.0)
    # x component AS WRITTEN at boundary.c:69: (1-c)*(-ky*ky + kz*kz*x + ...) with s*(-kz*y)
    assert np.allclose(u[:, 0], -X[:, 1] + (0.0 + 1.0 * X[:, 0]))


@pytest.mark.parametrize("problem,incs", [("linElas", 1), ("hyperFS", 2)])
def test_solver_converges_on_oracle(oracle, problem, incs):
    mesh = hollow_cylinder_mesh(1, 6, 2, z0=-1.0, z1=1.0)
    p = SolidProblem(oracle, mesh, 2, problem, nu=0.3, E=10.0, bc_sides=[998, 999])
    s = NewtonPMG(p, clamp=CLAMP)
    st = s.solve(incs)
    assert st.converged and st.increments == incs
    u = s.U.to_numpy()
    # converged state: residual with the final boundary values is small relative to the first residual
    first = [h[4] for h in st.history if h[0] == incs][0]
    assert st.history[-1][4] < 1e-6 * max(first, 1e-300) or st.history[-1][4] < 1e-9
    assert np.all(u[p.levels[p.fine].mask != 0] == 0.0)          # L-layout: constrained entries stay zero
    if problem == "linElas":
        assert st.newton_its == 1                                   # linear problem: one Newton step


def test_convergence_in_the_last_allowed_newton_iteration(oracle):
    """A step that meets snes_rtol in iteration snes_maxit has converged (the tolerance test sits at the top of the
    loop, so the flag is decided on the residual afterwards); one iteration fewer has not."""
    mesh = hollow_cylinder_mesh(1, 6, 2, z0=-1.0, z1=1.0)
    p = SolidProblem(oracle, mesh, 2, "hyperFS", nu=0.3, E=10.0, bc_sides=[998, 999])
    ref = NewtonPMG(p, clamp=CLAMP)
    st = ref.solve(2)
    per_inc = [sum(1 for h in st.history if h[0] == inc) for inc in (1, 2)]
    need = max(per_inc)
    assert st.converged and need >= 2
    s = NewtonPMG(p, clamp=CLAMP, snes_maxit=need)
    st2 = s.solve(2)
    assert st2.converged and st2.increments == 2 and st2.newton_its == st.newton_its
    assert np.array_equal(s.U.to_numpy(), ref.U.to_numpy())
    s = NewtonPMG(p, clamp=CLAMP, snes_maxit=need - 1)
    st3 = s.solve(2)
    assert not st3.converged


@pytest.mark.gpu
def test_device_solve_matches_oracle_solve(oracle, gpu):
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    sols = []
    for c in (oracle, gpu):
        p = SolidProblem(c, mesh, 2, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
        s = NewtonPMG(p, clamp=CLAMP)
        st = s.solve(2)
        assert st.converged
        sols.append((s.U.to_numpy(), st.newton_its))
    assert sols[0][1] == sols[1][1]
    assert rel_err(sols[1][0], sols[0][0]) < 1e-8


@pytest.mark.gpu
def test_config3_full_solve_on_the_device(gpu):
    """BASELINE config 3 as stated: hyperSS, cylinder8_5580e_4ss_us (the reference's own mesh), degree 4, the FULL
    Newton-CG-pMG solve (10 load increments, levels p = 1, 2, 4, assembled coarse level) on one MI355X.  Pinned: convergence
    of every increment, the Newton count (3 per increment), a Krylov count in the band the matrix-free and the assembled
    coarse solve both give, the clamp displacement reached, and a final residual at the solver's tolerance.
    THE LOAD the pinned counts belong to: -bc_clamp_998_translate 0,-0.05,0.1 -- a TENTH of the README's 0,-0.5,1
    (README.rst:63, whose minimal command runs linElas); the README's load itself: test_config3_readme_load_on_the_device."""
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_5580e_4ss_us.npz"))
    p = SolidProblem(gpu, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
    assert p.degrees == [1, 2, 4] and p.n_free() == 1150068
    tr = (0.0, -0.05, 0.1)
    s = NewtonPMG(p, clamp={998: dict(translate=tr), 999: dict()}, coarse="assembled", coarse_cheb_its=40, coarse_cheb_ratio=100.0, graph=True)
    st = s.solve(10)
    assert st.converged and st.increments == 10
    assert st.newton_its == 30
    assert 400 <= st.ksp_its <= 1400, st.ksp_its
    last = [h for h in st.history if h[0] == 10]
    first_of_last = last[0][4]
    assert st.history[-1][4] < 1e-6 * max(1.0, first_of_last) or st.history[-1][4] < 1e-8
    u = s.U.to_numpy()
    assert np.all(u[p.levels[p.fine].mask != 0] == 0.0)      # L-layout: constrained entries stay zero (the clamp values live in Xloc)
    umax = np.abs(u.reshape(-1, 3)).max(axis=0)
    # the free nodes next to the translated clamp (0, -0.05, 0.1) follow it (recorded on the device: 0.0031, 0.0515, 0.0998)
    assert 0.09 < umax[2] <= 0.1 + 1e-9 and 0.045 < umax[1] < 0.06 and umax[0] < 0.01, umax


def test_fused_apply_entries_on_the_oracle_are_their_two_steps(oracle):
    """CeedXOperatorApplyChebyshev / ApplyResidual as the oracle restates them: CeedOperatorApply, then the vector update."""
    import ctypes as C
    from ceedpetscsolid_amd.mesh import box_mesh
    p = SolidProblem(oracle, box_mesh(2, 2, 2), 2, "linElas", nu=0.3, E=1.0, bc_sides=[1])
    L, op, n = oracle.L, p.levels[p.fine].opJacob, p.lsize()
    rng = np.random.default_rng(0)
    free = (p.levels[p.fine].mask == 0).astype(np.float64)
    a = {k: rng.uniform(-1, 1, n) * free for k in ("x", "d", "r", "b", "dinv")}
    for first in (False, True):
        res = []
        for fused in (False, True):
            v = {k: oracle.vector(n).set_array(a[k]) for k in a}
            t = oracle.vector(n)
            src = v["x"] if first else v["d"]
            c1, c2 = C.c_double(0.4), C.c_double(0.0 if first else 0.3)
            if fused:
                L.chk(L.lib.CeedXOperatorApplyChebyshev(op.h, src.h, t.h, v["x"].h, v["d"].h, v["r"].h, v["b"].h if first else None, v["dinv"].h, c1, c2, 0))
            else:
                op.apply(src, t)
                if first:
                    L.chk(L.lib.CeedXVectorChebyshevStart(v["x"].h, v["d"].h, v["r"].h, v["b"].h, t.h, v["dinv"].h, c1, 0))
                else:
                    L.chk(L.lib.CeedXVectorChebyshevUpdate(v["x"].h, v["d"].h, v["r"].h, t.h, v["dinv"].h, c1, c2, 0))
            res.append([v[k].to_numpy() for k in ("x", "d", "r")])
        for u, w in zip(*res):
            assert np.array_equal(u, w)
    X, B, T, W = oracle.vector(n).set_array(a["x"]), oracle.vector(n).set_array(a["b"]), oracle.vector(n), oracle.vector(n)
    L.chk(L.lib.CeedXOperatorApplyResidual(op.h, X.h, T.h, B.h, W.h))
    Y = oracle.vector(n); op.apply(X, Y)
    assert np.array_equal(W.to_numpy(), a["b"] - Y.to_numpy())
    p.destroy()


def test_auto_form_on_a_backend_without_graph_capture(oracle):
    """graph="auto" on the CPU oracle (which cannot record): the measurement covers the eager forms only and the solve goes on."""
    from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh
    p = SolidProblem(oracle, hollow_cylinder_mesh(1, 6, 4, z0=-1.0, z1=1.0), 2, "hyperSS", nu=0.3, E=10.0, bc_sides=[998, 999])
    ref = NewtonPMG(p, clamp=CLAMP, coarse="amg", graph=False, fuse_epilogue=False)
    st0 = ref.solve(1)
    s = NewtonPMG(p, clamp=CLAMP, coarse="amg", graph="auto", fuse_epilogue="auto")
    st = s.solve(1)
    assert st.converged and s.graph is False and set(s.tuning["vcycle_ms"]) == {"fused+eager", "two_pass+eager"}
    assert (st.newton_its, st.ksp_its) == (st0.newton_its, st0.ksp_its)
    p.destroy()


@pytest.mark.gpu
def test_vcycle_form_is_chosen_by_measurement_and_does_not_change_the_solve(gpu):
    """graph="auto", fuse_epilogue="auto": the first Newton step times the V-cycle eager / replayed and with the smoother's
    Chebyshev step fused into the apply's epilogue / as a pass of its own, and keeps the fastest.  The four forms give the same
    bits, so the Newton and Krylov counts and the solution equal those of the explicit eager two-pass solve."""
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_5580e_4ss_us.npz"))     # config 3 at its pinned load
    out = []
    for kw in (dict(graph=False, fuse_epilogue=False), dict(graph="auto", fuse_epilogue="auto"), dict(graph=True, fuse_epilogue=True)):
        p = SolidProblem(gpu, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
        s = NewtonPMG(p, clamp=CLAMP, coarse="amg", **kw)
        st = s.solve(10)
        assert st.converged
        out.append((st.newton_its, st.ksp_its, s.U.to_numpy(), s.tuning))
        p.destroy()
    assert out[1][3] is not None and len(out[1][3]["vcycle_ms"]) == 4 and out[0][3] is None
    for o in out[1:]:
        assert (o[0], o[1]) == (out[0][0], out[0][1])
        assert np.array_equal(o[2], out[0][2])


@pytest.mark.gpu
def test_config3_readme_load_on_the_device(gpu):
    """BASELINE config 3 at the load SURVEY 8(d) states -- -bc_clamp 998,999 -bc_clamp_998_translate 0,-0.5,1 (README.rst:63) -- with
    hyperSS at degree 4 on the reference's own mesh.  What decides convergence is the size of a load increment, not the line search
    (profiles/r04_config3_readme_load.txt: the first increment of 10, 20 or 40 inverts the first layer of degree-4 elements at the clamp
    -- det F < 0, tr eps < -1 -- whichever of the three line searches runs, a never-rejecting one-step CP search in PETSc's form (restated from memory, unverified: profiles/r05_ab_experiments.txt item 9) included; a
    hundredth of the translation per increment converges with all three).  With -num_steps 100 the whole solve converges: recorded on
    the device 300 Newton / 4 929 Krylov iterations, clamp displacement reached (0.0316, 0.5155, 0.9984)."""
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_5580e_4ss_us.npz"))
    p = SolidProblem(gpu, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
    s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.5, 1.0)), 999: dict()}, coarse="amg", graph=True)
    st = s.solve(100)
    assert st.converged and st.increments == 100
    assert 280 <= st.newton_its <= 330, st.newton_its
    assert 4000 <= st.ksp_its <= 6000, st.ksp_its
    umax = np.abs(s.U.to_numpy().reshape(-1, 3)).max(axis=0)
    assert 0.99 < umax[2] <= 1.0 + 1e-9 and 0.5 < umax[1] < 0.53 and umax[0] < 0.05, umax
    assert st.history[-1][4] < 1e-6


def test_chebyshev_coarse_solver_converges_on_oracle(oracle):
    mesh = hollow_cylinder_mesh(1, 6, 2, z0=-1.0, z1=1.0)
    p = SolidProblem(oracle, mesh, 2, "hyperSS", nu=0.3, E=10.0, bc_sides=[998, 999])
    ref = NewtonPMG(p, clamp=CLAMP)
    assert ref.solve(1).converged
    s = NewtonPMG(p, clamp=CLAMP, coarse="chebyshev", coarse_cheb_its=20, coarse_cheb_ratio=50.0)
    assert s.solve(1).converged
    assert rel_err(s.U.to_numpy(), ref.U.to_numpy()) < 1e-8
    with pytest.raises(ValueError):
        NewtonPMG(p, clamp=CLAMP, coarse="cg", graph=True)


@pytest.mark.gpu
def test_graph_replayed_vcycle_matches_eager_vcycle(gpu):
    """CeedXGraph*: a recorded V-cycle replays the same kernels on the same data, so the solve's
    iterates match the eager solve; host-needing calls are refused while recording."""
    from ceedpetscsolid_amd import ceed as cd
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    out = []
    for graph in (False, True):
        p = SolidProblem(gpu, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
        s = NewtonPMG(p, clamp=CLAMP, coarse="chebyshev", coarse_cheb_its=20, coarse_cheb_ratio=50.0, graph=graph)
        st = s.solve(2)
        assert st.converged
        out.append((s.U.to_numpy(), st.newton_its, st.ksp_its, st.jacobian_applies))
    # every scatter in the V-cycle (Jacobian, Prolong_Ceed, Restrict_Ceed) is a deterministic E-vector +
    # per-node sum, so the replayed solve is BITWISE the eager one
    assert out[0][1:3] == out[1][1:3]                 # Newton and Krylov iteration counts
    assert np.array_equal(out[0][0], out[1][0])
    x, y = gpu.vector(8).set_value(1.0), gpu.vector(8).set_value(2.0)

    def bad():
        x.dot(y) if hasattr(x, "dot") else gpu.L.chk(gpu.L.lib.CeedXVectorDot(x.h, y.h, None, cd.C.byref(cd.C.c_double())))
    with pytest.raises(cd.CeedError):
        gpu.capture(bad)
    x.set_value(3.0)                       # the Ceed is usable again after the refused recording
    assert np.all(x.to_numpy() == 3.0)


@pytest.mark.parametrize("problem", ["linElas", "hyperFS"])
def test_assembled_coarse_matrix_equals_matrix_free_operator(oracle, problem):
    """the p=1 matrix assembled from element matrices (assembly.py, CeedXCsr*) acts like the
    matrix-free coarse Jacobian on the free dofs and like the identity on the constrained ones."""
    from ceedpetscsolid_amd.assembly import AssembledLevel
    mesh = hollow_cylinder_mesh(1, 6, 2, z0=-1.0, z1=1.0)
    p = SolidProblem(oracle, mesh, 2, problem, nu=0.3, E=10.0, bc_sides=[998])
    n = p.lsize()
    X, R = oracle.vector(n), oracle.vector(n)
    X.set_array(p.smooth_state(0.05) * (p.levels[p.fine].mask == 0)); p.form_residual(X, R)
    A = AssembledLevel(p, 0); A.assemble()
    n0 = p.lsize(0)
    free = p.levels[0].mask == 0
    x = np.random.default_rng(1).uniform(-1, 1, n0)
    xv, y1, y2, d1, d2 = (oracle.vector(n0) for _ in range(5))
    xv.set_array(x)
    p.apply_jacobian(0, xv, y1); A.apply(xv, y2)
    a, b = y1.to_numpy(), y2.to_numpy()
    assert rel_err(b[free], a[free]) < 1e-13 and np.array_equal(b[~free], x[~free])
    p.get_diag(0, d1); A.diagonal(d2)
    assert rel_err(d2.to_numpy()[free], d1.to_numpy()[free]) < 1e-13 and np.all(d2.to_numpy()[~free] == 1.0)
    # symmetric (the tangent of a hyperelastic energy), 81-point stencil at most
    assert A.nnz <= 81 * n0


def test_solver_with_assembled_coarse_level_converges_on_oracle(oracle):
    mesh = hollow_cylinder_mesh(1, 6, 2, z0=-1.0, z1=1.0)
    p = SolidProblem(oracle, mesh, 2, "hyperSS", nu=0.3, E=10.0, bc_sides=[998, 999])
    ref = NewtonPMG(p, clamp=CLAMP, coarse="chebyshev", coarse_cheb_its=20, coarse_cheb_ratio=50.0)
    assert ref.solve(1).converged
    s = NewtonPMG(p, clamp=CLAMP, coarse="assembled", coarse_cheb_its=20, coarse_cheb_ratio=50.0)
    st = s.solve(1)
    assert st.converged and st.coarse_spmv > 0
    assert st.ksp_its == ref.stats.ksp_its            # the same polynomial of the same operator
    assert rel_err(s.U.to_numpy(), ref.U.to_numpy()) < 1e-8


@pytest.mark.gpu
def test_assembled_coarse_level_on_device(oracle, gpu):
    """Device assembly == oracle assembly (same element-matrix entries, same summation order), and the
    graph-replayed solve with the assembled coarse level converges to the eager matrix-free answer."""
    from ceedpetscsolid_amd.assembly import AssembledLevel
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    ys = []
    for c in (oracle, gpu):
        p = SolidProblem(c, mesh, 4, "hyperFS", nu=0.3, E=1e3, bc_sides=[998, 999])
        n = p.lsize()
        X, R = c.vector(n), c.vector(n)
        X.set_array(p.smooth_state(0.02) * (p.levels[p.fine].mask == 0)); p.form_residual(X, R)
        A = AssembledLevel(p, 0); A.assemble()
        n0 = p.lsize(0)
        x = c.vector(n0).set_array(np.random.default_rng(3).uniform(-1, 1, n0))
        y, ymf = c.vector(n0), c.vector(n0)
        A.apply(x, y); p.apply_jacobian(0, x, ymf)
        free = p.levels[0].mask == 0
        assert rel_err(y.to_numpy()[free], ymf.to_numpy()[free]) < 1e-12
        ys.append(y.to_numpy())
    assert rel_err(ys[1], ys[0]) < 1e-12
    p = SolidProblem(gpu, mesh, 2, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
    ref = NewtonPMG(p, clamp=CLAMP, coarse="chebyshev", coarse_cheb_its=20, coarse_cheb_ratio=50.0)
    assert ref.solve(2).converged
    s = NewtonPMG(p, clamp=CLAMP, coarse="assembled", coarse_cheb_its=20, coarse_cheb_ratio=50.0, graph=True)
    assert s.solve(2).converged
    assert rel_err(s.U.to_numpy(), ref.U.to_numpy()) < 1e-8


def vector_helpers(ceed, tol):
    """CeedXVectorChebyshevStart / WAXPBY / DotTo / ScalarDivide / AXPBYScalars against numpy."""
    import ctypes as C
    L = ceed.L
    rng = np.random.default_rng(11)
    n = 1000
    a = {k: rng.standard_normal(n) for k in ("x", "d", "r", "b", "t", "dinv", "y")}
    v = {k: ceed.vector(n).set_array(val.copy()) for k, val in a.items()}
    L.chk(L.lib.CeedXVectorChebyshevStart(v["x"].h, v["d"].h, v["r"].h, v["b"].h, v["t"].h, v["dinv"].h, C.c_double(0.7), 0))
    r = a["b"] - a["t"]; d = 0.7 * a["dinv"] * r
    assert np.abs(v["r"].to_numpy() - r).max() < tol and np.abs(v["d"].to_numpy() - d).max() < tol
    assert np.abs(v["x"].to_numpy() - (a["x"] + d)).max() < tol
    L.chk(L.lib.CeedXVectorChebyshevStart(v["x"].h, v["d"].h, v["r"].h, v["b"].h, None, v["dinv"].h, C.c_double(0.7), 1))
    assert np.abs(v["r"].to_numpy() - a["b"]).max() < tol and np.abs(v["x"].to_numpy() - 0.7 * a["dinv"] * a["b"]).max() < tol
    with pytest.raises(Exception):
        L.chk(L.lib.CeedXVectorChebyshevStart(v["x"].h, v["d"].h, v["r"].h, v["r"].h, None, v["dinv"].h, C.c_double(0.7), 1))
    w = ceed.vector(n)
    L.chk(L.lib.CeedXVectorWAXPBY(w.h, C.c_double(2.0), v["b"].h, C.c_double(-3.0), v["t"].h))
    assert np.abs(w.to_numpy() - (2.0 * a["b"] - 3.0 * a["t"])).max() < tol
    sc = ceed.vector(8).set_value(0.0)
    L.chk(L.lib.CeedXVectorDotTo(v["b"].h, v["t"].h, None, sc.h, 2))
    L.chk(L.lib.CeedXVectorDotTo(v["b"].h, v["b"].h, v["dinv"].h, sc.h, 3))
    L.chk(L.lib.CeedXScalarDivide(sc.h, 4, 2, 3, C.c_double(-2.0)))
    L.chk(L.lib.CeedXScalarDivide(sc.h, 5, 2, -1, C.c_double(1.0)))
    L.chk(L.lib.CeedXScalarDivide(sc.h, 6, 2, 7, C.c_double(1.0)))          # zero denominator: 0, not inf
    s = sc.to_numpy()
    bt, bwb = a["b"] @ a["t"], (a["dinv"] * a["b"]) @ a["b"]
    assert abs(s[2] - bt) < 100 * tol and abs(s[3] - bwb) < 100 * tol
    assert abs(s[4] - (-2.0 * bt / bwb if bwb > 0 else 0.0)) < 100 * tol and s[5] == s[2] and s[6] == 0.0
    L.chk(L.lib.CeedXVectorAXPBYScalars(v["y"].h, sc.h, 2, C.c_double(-1.0), v["b"].h, -1, C.c_double(0.5)))
    assert np.abs(v["y"].to_numpy() - (-s[2] * a["b"] + 0.5 * a["y"])).max() < 1e3 * tol
    with pytest.raises(Exception):
        L.chk(L.lib.CeedXVectorDotTo(v["b"].h, v["t"].h, None, sc.h, 8))


def test_vector_helpers_on_oracle(oracle):
    vector_helpers(oracle, 1e-14)


@pytest.mark.gpu
def test_vector_helpers_on_device(gpu):
    vector_helpers(gpu, 1e-13)


def lanczos_paths_agree(ceed, tol):
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    p = SolidProblem(ceed, mesh, 2, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
    s = NewtonPMG(p, clamp=CLAMP, coarse="assembled")
    s.U.set_value(0.0); s._set(s.bcv, s.bc_values(0.5)); s.residual(s.U, s.R)
    s.setup_preconditioner()
    for lv in range(s.nlev):
        ah, bh = s._lanczos_host(lv, 10)
        ad, bd = s._lanczos_device(lv, 10)
        assert len(ah) == len(ad) == 10
        assert rel_err(np.array(ad), np.array(ah)) < tol and rel_err(np.array(bd), np.array(bh)) < tol
        assert 1.0 < s.emax[lv] < 10.0
    # the reciprocal of the diagonal never left the device: zeros (constrained rows) stay zeros, the rest is positive
    for lv in range(s.nlev):
        d = s.w[lv]["dinv"].to_numpy(); m = p.levels[lv].mask != 0
        assert np.all(d[m] == 0.0) and np.all(d[~m] > 0.0)


def test_eigenvalue_estimate_with_device_scalars_equals_host_path_on_oracle(oracle):
    lanczos_paths_agree(oracle, 1e-15)


@pytest.mark.gpu
def test_eigenvalue_estimate_with_device_scalars_equals_host_path_on_device(gpu):
    lanczos_paths_agree(gpu, 1e-12)


@pytest.mark.gpu
def test_solver_interface_sums_and_dots_go_through_the_library(product_lib):
    """ONE halo mechanism for everything.  With the library's exchange (halo.RcclHalo per level) the
    solver's operator outputs -- Jacobian, residual, transfers, diagonal -- are summed over the interface by CeedXHalo* and its
    dots by CeedXCommAllReduce, nothing through torch.  On ONE GPU: an emulated rank of a 4-rank cylinder (real sub-mesh
    and neighbour lists, the exchange sent to the rank itself), so every sum is the vector folded onto itself -- which is
    exactly predictable: y + sum over the neighbour lists of y[list].  The split-phase Jacobian with the exchange under the
    interior elements, and the recorded V-cycle (RCCL sends inside a hipGraph), must reproduce the eager, unsplit numbers."""
    import torch
    from ceedpetscsolid_amd import ceed as cd
    from ceedpetscsolid_amd.halo import HaloExchange, RcclHalo, interface_elements, part_cylinder, virtual_world
    from ceedpetscsolid_amd.mesh import reorder_elements_first
    K, N = 1, 4
    part = lambda r: part_cylinder(r, N, 2, 12, 12)
    mesh = part(K)
    lead = interface_elements(mesh, virtual=virtual_world(K, N, mesh, part, 1))
    mesh = reorder_elements_first(mesh, lead)
    ceed = cd.Ceed(product_lib, "/gpu/hip/mi355x")
    ceed.set_stream(torch.cuda.current_stream().cuda_stream)
    p = SolidProblem(ceed, mesh, 2, "hyperSS", nu=0.3, E=10.0, bc_sides=[])
    halos = [HaloExchange(mesh, lv.dofmap, device="cuda", virtual=virtual_world(K, N, mesh, part, deg)) for lv, deg in zip(p.levels, p.degrees)]
    rh = [RcclHalo(ceed, h, emulate_self=True) for h in halos]
    s = NewtonPMG(p, halo=halos, rccl=rh, lead_elements=int(lead.sum()), coarse="chebyshev", coarse_cheb_its=10, graph=True)
    assert s._split and s.rhalos == rh
    rng = np.random.default_rng(4)
    s.U.set_value(0.0); s.residual(s.U, s.R)

    def folded(y, h):
        out = y.copy()
        for nb in h.neigh:
            i = nb.dof_idx.cpu().numpy()
            out[i] += y[i]
        return out
    for lv in range(s.nlev):
        n = p.lsize(lv)
        x, y, y0 = s._vec(n, lv), s._vec(n, lv), ceed.vector(n)
        s._set(x, rng.uniform(-1, 1, n))
        s.A(lv, x, y)                                   # split-phase + exchange in the library
        p.apply_jacobian(lv, x, y0)                     # plain apply
        assert np.array_equal(y.to_numpy(), folded(y0.to_numpy(), halos[lv])), lv
        # dots: owner-weighted, summed over the (one) rank on the device
        w = halos[lv].owner_weight * (p.levels[lv].mask == 0)
        assert abs(s.dot(x, y, lv=lv) - float((x.to_numpy() * y.to_numpy() * w).sum())) < 1e-9 * abs(float((x.to_numpy() * y.to_numpy() * w).sum())) + 1e-300
    # the V-cycle, eager and recorded (its halo sums are RCCL sends and receives inside the graph: in order on the
    # capturing stream this RCCL records them correctly, tools/rccl_capture_probe.py)
    top = s.nlev - 1
    s.setup_preconditioner()
    r, z, z2 = s.w[top]["b"], s.kz, s._vec(p.lsize(), top)
    s._set(r, rng.uniform(-1, 1, p.lsize()) * (p.levels[top].mask == 0))
    s.vcycle(top, r, z2)
    want = z2.to_numpy().copy()
    assert np.isfinite(want).all()
    s.record_preconditioner(r, z)
    assert s._pc_graph is not None
    for _ in range(2):
        z.set_value(3.0)
        s.precondition(r, z)
        assert np.array_equal(z.to_numpy(), want)
    for h in rh:
        h.destroy()
