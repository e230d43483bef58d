"""1-D rules and basis tables of BOTH libraries against tests/golden/basis_tables.npz -- 50-digit values computed from
first principles by oracle/gen_tables_golden.py, independent of either library's generator (reference call
sites src/setuplibceed.c:335-347, 782-803).  The product's table generator (csrc/ceed_basis.cpp) and the oracle's
(oracle_ceed.c) are thereby pinned separately: a bug shared by the two would fail here."""
import ctypes as C
import os

import numpy as np
import pytest

from ceedpetscsolid_amd import ceed as cd

from conftest import GOLDEN

TOL = 1e-14
_G = np.load(os.path.join(GOLDEN, "basis_tables.npz"))
GAUSS_PAIRS = sorted({tuple(map(int, k.split("_")[2:])) for k in _G.files if k.startswith("interp_gauss_")})
CTOF_PAIRS = sorted({tuple(map(int, k.split("_")[2:])) for k in _G.files if k.startswith("interp_lobatto_")})
LIBS = [pytest.param("oracle", id="oracle"), pytest.param("product", id="product", marks=pytest.mark.gpu)]


def _ceed(request, which):
    return request.getfixturevalue("oracle" if which == "oracle" else "gpu")


def _rule(L, fn, Q):
    x, w = np.zeros(Q), np.zeros(Q)
    dp = C.POINTER(C.c_double)
    L.chk(getattr(L.lib, fn)(Q, x.ctypes.data_as(dp), w.ctypes.data_as(dp)))
    return x, w


@pytest.mark.parametrize("which", LIBS)
def test_quadrature_rules_match_the_golden_values(request, which):
    L = _ceed(request, which).L
    for Q in range(1, 9):
        x, w = _rule(L, "CeedGaussQuadrature", Q)
        assert np.abs(x - _G[f"gauss_x_{Q}"]).max() < TOL and np.abs(w - _G[f"gauss_w_{Q}"]).max() < TOL, Q
    for Q in range(2, 9):
        x, w = _rule(L, "CeedLobattoQuadrature", Q)
        assert np.abs(x - _G[f"lobatto_x_{Q}"]).max() < TOL and np.abs(w - _G[f"lobatto_w_{Q}"]).max() < TOL, Q


@pytest.mark.parametrize("which", LIBS)
@pytest.mark.parametrize("P,Q", GAUSS_PAIRS)
def test_gauss_basis_tables_match_the_golden_values(request, which, P, Q):
    """basisu / basisx / level bases: P Lobatto nodes -> Q Gauss points (setuplibceed.c:335-341, 782-784)."""
    b = _ceed(request, which).basis_lagrange(3, 3, P, Q, cd.GAUSS)
    assert np.abs(b.interp1d - _G[f"interp_gauss_{P}_{Q}"]).max() < TOL
    assert np.abs(b.grad1d - _G[f"grad_gauss_{P}_{Q}"]).max() < TOL * 10      # entries up to ~10
    assert np.abs(b.qweight1d - _G[f"gauss_w_{Q}"]).max() < TOL
    b.destroy()


@pytest.mark.parametrize("which", LIBS)
@pytest.mark.parametrize("P,Q", CTOF_PAIRS)
def test_lobatto_basis_tables_match_the_golden_values(request, which, P, Q):
    """basisCtoF (coarse nodes -> fine nodes, setuplibceed.c:799-803) and basisDiagnostic (:347-348)."""
    b = _ceed(request, which).basis_lagrange(3, 3, P, Q, cd.GAUSS_LOBATTO)
    assert np.abs(b.interp1d - _G[f"interp_lobatto_{P}_{Q}"]).max() < TOL
    assert np.abs(b.grad1d - _G[f"grad_lobatto_{P}_{Q}"]).max() < TOL * 30    # entries up to ~30 at P = 8 (end points)
    assert np.abs(b.qweight1d - _G[f"lobatto_w_{Q}"]).max() < TOL
    b.destroy()
