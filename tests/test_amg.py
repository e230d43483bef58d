"""The coarse solve of the p-multigrid cycle as ONE cycle of an aggregation hierarchy on the assembled matrix (the role of
PCGAMG under KSPPREONLY, elasticity.c:568-585): the library pieces (`CeedXCsrCreateRect`, `CeedXCsrCreateProduct`,
`CeedXCsrUpdate`, `CeedXCsrInvertDenseSPD`) against scipy / numpy on the CPU oracle, the device against the oracle, and the
solve with it against the solve with the Chebyshev coarse solver."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.amg import AggregationAMG, aggregate_nodes, rigid_body_prolongation
from ceedpetscsolid_amd.assembly import AssembledLevel
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, load_mesh_npz
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG
from conftest import GOLDEN, rel_err

CLAMP = {998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}


def to_scipy(m: cd.Csr) -> sp.csr_matrix:
    nr, nc, nz, rp, cl = m.pattern()
    return sp.csr_matrix((m.values(), cl, rp), shape=(nr, nc))


def random_pieces(ceed, n=57, nc=13, seed=3):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=0.15, random_state=seed, format="csr")
    A = (A + A.T + sp.diags(np.full(n, 4.0))).tocsr(); A.sort_indices()
    P = sp.random(n, nc, density=0.3, random_state=seed + 1, format="csr"); P.sort_indices()
    Pt = P.T.tocsr(); Pt.sort_indices()
    # the variable operand: an assembled matrix whose COO entries are its own entries
    a = cd.Csr(ceed, A.indptr, A.indices, np.arange(A.nnz))
    return rng, A, P, Pt, a


def products_follow_the_variable_operand(ceed, tol):
    rng, A, P, Pt, a = random_pieces(ceed)
    n, nc = P.shape
    p, pt = cd.Csr.rect(ceed, n, nc, P.indptr, P.indices, P.data), cd.Csr.rect(ceed, nc, n, Pt.indptr, Pt.indices, Pt.data)
    T = cd.Csr.product(a, p, variable=0)
    Ac = cd.Csr.product(pt, T, variable=1, dense=True)
    assert (Ac.nrows, Ac.ncols, Ac.nnz) == (nc, nc, nc * nc)
    for trial in range(2):                       # the values change, the patterns stay
        vals = A.data * (1.0 + 0.1 * trial * rng.standard_normal(A.nnz))
        a.assemble(ceed.vector(A.nnz).set_array(vals))
        T.update(); Ac.update()
        Av = sp.csr_matrix((vals, A.indices, A.indptr), shape=A.shape)
        assert abs(to_scipy(T) - Av @ P).max() < tol
        assert np.abs(to_scipy(Ac).toarray() - (P.T @ Av @ P).toarray()).max() < tol
    # rectangular apply: x of ncols, y of nrows
    x = rng.standard_normal(nc)
    y = ceed.vector(n)
    p.apply(ceed.vector(nc).set_array(x), y)
    assert np.abs(y.to_numpy() - P @ x).max() < tol
    with pytest.raises(cd.CeedError):
        cd.Csr.product(p, p, variable=0)         # same matrix twice / shapes do not chain


def dense_inverse(ceed, n, tol):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    S = B @ B.T + n * np.eye(n)
    D = cd.Csr.rect(ceed, n, n, np.arange(n + 1) * n, np.tile(np.arange(n), n), S.reshape(-1))
    D.invert_dense_spd()
    inv = D.values().reshape(n, n)
    assert np.abs(inv @ S - np.eye(n)).max() < tol
    assert np.array_equal(inv, inv.T)
    # applied like any other matrix
    x = rng.standard_normal(n)
    y = ceed.vector(n)
    D.apply(ceed.vector(n).set_array(x), y)
    assert rel_err(y.to_numpy(), np.linalg.solve(S, x)) < tol
    # not positive definite: refused loudly; a sparse pattern: refused
    S[n // 2, n // 2] = -1.0
    bad = cd.Csr.rect(ceed, n, n, np.arange(n + 1) * n, np.tile(np.arange(n), n), S.reshape(-1))
    with pytest.raises(cd.CeedError):
        bad.invert_dense_spd()
    if n > 1:
        eye = cd.Csr.rect(ceed, n, n, np.arange(n + 1), np.arange(n), np.ones(n))
        with pytest.raises(cd.CeedError):
            eye.invert_dense_spd()


def test_products_follow_the_variable_operand_on_oracle(oracle):
    products_follow_the_variable_operand(oracle, 1e-13)


@pytest.mark.parametrize("n", [1, 31, 32, 70])
def test_dense_inverse_on_oracle(oracle, n):
    dense_inverse(oracle, n, 1e-11)


def test_aggregates_cover_the_graph_and_rigid_body_modes_are_reproduced():
    mesh = hollow_cylinder_mesh(2, 8, 6)
    from ceedpetscsolid_amd.mesh import build_dofmap
    dm = build_dofmap(mesh, 1)
    # node graph of the p=1 mesh: nodes sharing an element
    off = dm.elem_nodes.astype(np.int64)
    r = np.repeat(off, 8, axis=1).reshape(-1); c = np.tile(off, (1, 8)).reshape(-1)
    G = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(dm.nnodes, dm.nnodes)).tocsr()
    agg, na = aggregate_nodes(G.indptr, G.indices)
    assert agg.min() == 0 and agg.max() == na - 1 and np.unique(agg).size == na
    sizes = np.bincount(agg)
    assert sizes.min() >= 4 and 8 < sizes.mean() < 60
    free = np.zeros(3 * dm.nnodes, dtype=bool)
    P0 = rigid_body_prolongation(agg, na, dm.node_coords, free)
    assert P0.shape == (3 * dm.nnodes, 6 * na)
    assert abs(P0.T @ P0 - sp.identity(6 * na)).max() < 1e-12          # orthonormal columns, disjoint supports
    # every global rigid-body motion lies in the range of P0
    X = dm.node_coords
    for u in (np.tile([1.0, 0.0, 0.0], dm.nnodes), np.stack([-X[:, 1], X[:, 0], 0 * X[:, 0]], axis=1).reshape(-1)):
        assert np.abs(P0 @ (P0.T @ u) - u).max() < 1e-12


def test_near_null_space_is_carried_down_the_levels():
    """B = P0 B1 = P0 P1 B2 exactly: the SVD factors of each aggregate's block split it into the prolongation's columns and
    the next level's near-null space, level after level (aggregates of aggregates)."""
    from ceedpetscsolid_amd.amg import rigid_body_modes, tentative_prolongation
    from ceedpetscsolid_amd.mesh import build_dofmap
    dm = build_dofmap(hollow_cylinder_mesh(2, 8, 6), 1)
    off = dm.elem_nodes.astype(np.int64)
    r = np.repeat(off, 8, axis=1).reshape(-1); c = np.tile(off, (1, 8)).reshape(-1)
    G = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(dm.nnodes, dm.nnodes)).tocsr()
    agg, na = aggregate_nodes(G.indptr, G.indices)
    con = np.zeros(3 * dm.nnodes, dtype=bool); con[:30] = True           # a few constrained dofs: their rows of B are zero
    B0 = rigid_body_modes(dm.node_coords, con)
    P0, B1, ptr1 = tentative_prolongation(agg, na, 3 * np.arange(dm.nnodes + 1), B0)
    assert ptr1.size == na + 1 and ptr1[-1] == P0.shape[1] == B1.shape[0]
    assert np.abs(P0 @ B1 - B0).max() < 1e-12
    agg2 = np.arange(na) // 4
    P1, B2, ptr2 = tentative_prolongation(agg2, int(agg2.max()) + 1, ptr1, B1)
    assert np.abs(P1 @ B2 - B1).max() < 1e-12 and np.abs(P0 @ (P1 @ B2) - B0).max() < 1e-12
    assert abs(P1.T @ P1 - sp.identity(P1.shape[1])).max() < 1e-12


def hierarchy_on_fixture(ceed, **kw):
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    p = SolidProblem(ceed, mesh, 2, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
    U, R = ceed.vector(p.lsize()), ceed.vector(p.lsize())
    X = p.levels[p.fine].dofmap.node_coords
    u = 0.02 * np.stack([np.sin(X[:, 1]) * X[:, 2], np.cos(X[:, 0]) * X[:, 2], np.sin(X[:, 0] + X[:, 1])], axis=1).reshape(-1)
    U.set_array(u * (p.levels[p.fine].mask == 0)); p.form_residual(U, R)
    a = AssembledLevel(p, 0); a.assemble()
    amg = AggregationAMG(a, **kw); amg.setup()
    return p, a, amg


def test_hierarchy_on_the_assembled_level_on_oracle(oracle):
    p, a, amg = hierarchy_on_fixture(oracle)
    A, P, Pt, T = to_scipy(a.csr), to_scipy(amg.P), to_scipy(amg.Pt), to_scipy(amg.T)
    assert abs(Pt - P.T).max() == 0.0
    assert abs(T - A @ P).max() < 1e-12 * abs(A).max()
    Ac = (P.T @ A @ P).toarray()
    inv = to_scipy(amg.Ac).toarray()
    assert np.abs(inv @ Ac - np.eye(amg.nc)).max() < 1e-10
    # constrained rows of the prolongation are empty: the correction leaves Dirichlet dofs alone
    con = p.levels[0].mask != 0
    assert abs(P[con]).sum() == 0.0
    # the coarse correction is the A-orthogonal projection onto range(P): a vector of the range is reproduced
    n = a.nrows
    xc = np.random.default_rng(5).standard_normal(amg.nc)
    r, z = oracle.vector(n).set_array(A @ (P @ xc)), oracle.vector(n)
    amg.restrict(r); amg.solve_coarsest(); amg.prolong(z)
    assert rel_err(z.to_numpy(), P @ xc) < 1e-9


def test_solve_with_the_aggregation_coarse_solve_on_oracle(oracle):
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    p = SolidProblem(oracle, mesh, 2, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
    ref = NewtonPMG(p, clamp=CLAMP, coarse="assembled", coarse_cheb_its=40, coarse_cheb_ratio=100.0)
    st_ref = ref.solve(1)
    s = NewtonPMG(p, clamp=CLAMP, coarse="amg")
    st = s.solve(1)
    assert st.converged and st_ref.converged and st.newton_its == st_ref.newton_its
    assert rel_err(s.U.to_numpy(), ref.U.to_numpy()) < 1e-7
    assert st.ksp_its < st_ref.ksp_its                  # a better coarse solve, at a fifth of the matrix products per cycle
    assert st.coarse_its < st_ref.coarse_its / 4


def test_three_levels_on_oracle(oracle):
    """A dense-level limit below the first coarse level's size adds a level: 3 000 -> 192 -> 18 rows on the fixture.  The
    second Galerkin matrix is P1^T A1 P1 of the FIRST one, its smoother bound is estimated per set-up, and the solve with
    the three-level cycle as coarse solver takes about as many iterations as with two levels."""
    p, a, amg = hierarchy_on_fixture(oracle, max_coarse_dofs=60)
    assert amg.info["levels"] == 3 and amg.info["rows"][1] == 192 and amg.info["rows"][2] <= 60
    l0, l1 = amg.levels
    A0, P0 = to_scipy(a.csr), to_scipy(l0.P)
    A1 = to_scipy(l0.Anext)
    assert abs(A1 - P0.T @ A0 @ P0).max() < 1e-10 * abs(A0).max()
    P1 = to_scipy(l1.P)
    A2 = (P1.T @ A1 @ P1).toarray()
    assert np.abs(to_scipy(l1.Anext).toarray() @ A2 - np.eye(l1.nc)).max() < 1e-9
    lam = np.linalg.eigvalsh(np.diag(1 / np.sqrt(A1.diagonal())) @ A1.toarray() @ np.diag(1 / np.sqrt(A1.diagonal()))).max()
    assert 0.7 * lam < l0.emax < 1.05 * lam                 # 10 Lanczos steps from a random start: a lower bound, close
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    its = []
    for mc in (1500, 60):
        pr = SolidProblem(oracle, mesh, 2, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
        s = NewtonPMG(pr, clamp=CLAMP, coarse="amg", amg_max_coarse_dofs=mc)
        st = s.solve(1)
        assert st.converged
        its.append(st.ksp_its)
    assert its[1] <= 1.3 * its[0]


# ---- the device against the oracle ------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_products_follow_the_variable_operand_on_device(gpu):
    products_follow_the_variable_operand(gpu, 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 31, 32, 33, 70, 1080])
def test_dense_inverse_on_device(gpu, n):
    dense_inverse(gpu, n, 1e-10)


@pytest.mark.gpu
def test_hierarchy_on_device_matches_oracle(oracle, gpu):
    (po, ao, mo), (pg, ag, mg) = hierarchy_on_fixture(oracle), hierarchy_on_fixture(gpu)
    assert mo.info["aggregates"] == mg.info["aggregates"] and mo.nc == mg.nc
    assert rel_err(mg.T.values(), mo.T.values()) < 1e-12
    # the inverses come from different algorithms (Gauss-Jordan on the device, Cholesky on the host)
    assert rel_err(mg.Ac.values(), mo.Ac.values()) < 1e-9
    n = ao.nrows
    r = np.random.default_rng(9).standard_normal(n) * (po.levels[0].mask == 0)
    out = []
    for c, m in ((oracle, mo), (gpu, mg)):
        z = c.vector(n)
        m.restrict(c.vector(n).set_array(r)); m.solve_coarsest(); m.prolong(z)
        out.append(z.to_numpy())
    assert rel_err(out[1], out[0]) < 1e-9


@pytest.mark.gpu
def test_three_level_cycle_on_device_matches_oracle(oracle, gpu):
    (po, ao, mo), (pg, ag, mg) = hierarchy_on_fixture(oracle, max_coarse_dofs=60), hierarchy_on_fixture(gpu, max_coarse_dofs=60)
    assert mo.info["rows"] == mg.info["rows"] and mo.info["levels"] == 3
    assert rel_err(mg.levels[0].Anext.values(), mo.levels[0].Anext.values()) < 1e-11
    assert abs(mg.levels[0].emax - mo.levels[0].emax) < 1e-9 * mo.levels[0].emax
    n = ao.nrows
    r = np.random.default_rng(9).standard_normal(n) * (po.levels[0].mask == 0)
    out = []
    for c, m in ((oracle, mo), (gpu, mg)):
        z = c.vector(n)
        m.restrict(c.vector(n).set_array(r)); m.solve_coarsest(); m.prolong(z)
        out.append(z.to_numpy())
    assert rel_err(out[1], out[0]) < 1e-8


@pytest.mark.gpu
def test_config3_solve_with_the_aggregation_coarse_solve(gpu):
    """BASELINE config 3 (hyperSS, cylinder8_5580e_4ss_us, degree 4, 10 increments) with coarse='amg', replayed as a graph:
    same Newton path and displacement as with the 40-step Chebyshev coarse solve, in well under half the Krylov iterations."""
    mesh = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_5580e_4ss_us.npz"))
    tr = (0.0, -0.05, 0.1)
    res = {}
    for coarse in ("assembled", "amg"):
        p = SolidProblem(gpu, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
        s = NewtonPMG(p, clamp={998: dict(translate=tr), 999: dict()}, coarse=coarse, graph=True)
        st = s.solve(10)
        assert st.converged and st.increments == 10 and st.newton_its == 30
        res[coarse] = (s.U.to_numpy(), st.ksp_its, st.seconds)
    assert rel_err(res["amg"][0], res["assembled"][0]) < 1e-6
    assert res["amg"][1] < 0.55 * res["assembled"][1], (res["amg"][1], res["assembled"][1])


@pytest.mark.gpu
def test_csr_apply_and_products_on_ragged_patterns(gpu):
    """CeedXCsrApply in the CSR-stream form (round 5: runs of rows swept into LDS) and CeedXCsrUpdate a wave per row, on patterns that
    reach every branch: empty rows, rows of one entry, runs cut by the 2 048-entry and the 256-row limits, a row LONGER than a run
    (3 000 entries: summed by the whole workgroup), product rows beyond the 4 096-entry LDS accumulator (the entry-per-lane kernel)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    n, m = 1500, 5000
    lens = rng.integers(0, 40, n)
    lens[7], lens[8], lens[400], lens[401:700] = 3000, 0, 2048, 1
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for k in lens]) if lens.sum() else np.zeros(0, dtype=np.int64)
    A = sp.csr_matrix((rng.uniform(-1, 1, rows.size), (rows, cols)), shape=(n, m))
    a = cd.Csr.rect(gpu, n, m, A.indptr, A.indices, A.data)
    x = rng.uniform(-1, 1, m)
    X, Y = gpu.vector(m).set_array(x), gpu.vector(n).set_value(7.0)
    a.apply(X, Y)
    ref = A @ x
    assert np.abs(Y.to_numpy() - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    assert Y.to_numpy()[8] == 0.0                                     # the empty row is written, not skipped
    # products: B (m x q) with a few very long rows makes rows of A B longer than 4 096 entries
    q = 9000
    blens = rng.integers(1, 6, m)
    blens[A.indices[A.indptr[7]:A.indptr[7] + 40]] = 150               # the long row of A meets long rows of B
    brow = np.repeat(np.arange(m), blens)
    bcol = np.concatenate([np.sort(rng.choice(q, k, replace=False)) for k in blens])
    B = sp.csr_matrix((rng.uniform(-1, 1, brow.size), (brow, bcol)), shape=(m, q))
    b = cd.Csr.rect(gpu, m, q, B.indptr, B.indices, B.data)
    C = cd.Csr.product(a, b, variable=0)
    C.update()
    nr, nc, nz, rp, cl = C.pattern()
    ref = (A @ B).tocsr(); ref.sort_indices()
    assert int(np.diff(rp).max()) > 4096 and nz == ref.nnz and np.array_equal(rp, ref.indptr) and np.array_equal(cl, ref.indices)
    assert np.abs(C.values(gpu) - ref.data).max() <= 1e-12 * np.abs(ref.data).max()
    # the same product with short rows only: the wave-per-row kernel
    A2 = A[20:390]
    a2 = cd.Csr.rect(gpu, A2.shape[0], m, A2.indptr, A2.indices, A2.data)
    C2 = cd.Csr.product(a2, b, variable=0)
    C2.update()
    ref2 = (A2 @ B).tocsr(); ref2.sort_indices()
    assert int(np.diff(C2.pattern()[3]).max()) <= 4096
    assert np.abs(C2.values(gpu) - ref2.data).max() <= 1e-12 * np.abs(ref2.data).max()
    for o in (C2, a2, C, b, a):
        o.destroy()
