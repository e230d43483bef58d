#!/usr/bin/env python3
"""Golden 1-D rules and basis tables, computed INDEPENDENTLY of both libraries (TEST INFRASTRUCTURE).

SURVEY 8(c) prescribes fixtures for the tables CeedBasisCreateTensorH1Lagrange builds (reference call sites
src/setuplibceed.c:335-347 basisu / basisx / basisEnergy / basisDiagnostic, :782-803 the level bases and the GLL CtoF
bases).  libCEED is absent, so the vectors come from first principles in 50-digit arithmetic (mpmath):

  * Gauss-Legendre points = roots of P_Q, weights 2 / ((1 - x^2) P_Q'(x)^2);
  * Gauss-Lobatto points = -1, roots of P_{Q-1}', +1, weights 2 / (Q (Q-1) P_{Q-1}(x)^2);
  * interp1d[q][p] = l_p(x_q), grad1d[q][p] = l_p'(x_q) for the Lagrange polynomials on the P Lobatto nodes, by the
    product formulas evaluated in 50 digits (no recurrence shared with oracle_ceed.c or ceed_api.cpp).

Output: tests/golden/basis_tables.npz (float64 roundings of the 50-digit values).  Both libraries -- the oracle and the
MI355X product -- are compared with it at 1e-14 (tests/test_basis_golden.py), which breaks the tie between their table
generators.

    python3 oracle/gen_tables_golden.py
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (P, Q) of SURVEY 8(c): Gauss rule; plus every level ladder's CtoF pair (P_coarse nodes -> P_fine Lobatto points)
GAUSS_PAIRS = [(2, 2), (2, 3), (3, 3), (2, 4), (3, 4), (4, 4), (2, 5), (3, 5), (5, 5), (2, 6), (3, 6), (5, 6), (6, 6),
               (2, 7), (3, 7), (5, 7), (7, 7), (2, 8), (3, 8), (5, 8), (8, 8)]
CTOF_PAIRS = [(2, 3), (3, 4), (3, 5), (5, 6), (5, 7), (5, 8), (2, 2), (3, 3), (4, 4), (5, 5), (7, 7), (4, 5), (6, 7), (7, 8)]


def legendre_roots(n, deriv=False):
    """Roots of P_n (or of P_n') in ascending order, polished by Newton in 50 digits from numpy's double roots."""
    c = np.polynomial.legendre.Legendre.basis(n)
    guess = np.sort((c.deriv() if deriv else c).roots().real)
    f = (lambda x: mp.diff(lambda t: mp.legendre(n, t), x)) if deriv else (lambda x: mp.legendre(n, x))
    roots = [mp.findroot(f, mp.mpf(float(g)), tol=mp.mpf(10) ** (-45), maxsteps=200) for g in guess]
    roots = [(r - s) / 2 for r, s in zip(roots, reversed(roots))]     # antisymmetrise: x_i = -x_{n-1-i}
    return roots


def gauss(Q):
    x = legendre_roots(Q)
    w = [2 / ((1 - xi ** 2) * mp.diff(lambda t: mp.legendre(Q, t), xi) ** 2) for xi in x]
    return x, w


def lobatto(Q):
    x = [mp.mpf(-1)] + (legendre_roots(Q - 1, deriv=True) if Q > 2 else []) + [mp.mpf(1)]
    w = [2 / (Q * (Q - 1) * mp.legendre(Q - 1, xi) ** 2) for xi in x]
    return x, w


def lagrange_tables(nodes, pts):
    P = len(nodes)
    B = [[None] * P for _ in pts]
    G = [[None] * P for _ in pts]
    for q, x in enumerate(pts):
        for j in range(P):
            den = mp.mpf(1)
            for m in range(P):
                if m != j:
                    den *= nodes[j] - nodes[m]
            val = mp.mpf(1)
            for m in range(P):
                if m != j:
                    val *= x - nodes[m]
            der = mp.mpf(0)
            for k in range(P):          # sum over the factor that is differentiated
                if k == j:
                    continue
                t = mp.mpf(1)
                for m in range(P):
                    if m != j and m != k:
                        t *= x - nodes[m]
                der += t
            B[q][j], G[q][j] = val / den, der / den
    return B, G


def f64(a):
    return np.array([[float(v) for v in row] for row in a]) if isinstance(a[0], list) else np.array([float(v) for v in a])


def main():
    out = {}
    for Q in range(1, 9):
        x, w = gauss(Q)
        assert abs(sum(w) - 2) < mp.mpf(10) ** (-40)
        out[f"gauss_x_{Q}"], out[f"gauss_w_{Q}"] = f64(x), f64(w)
    for Q in range(2, 9):
        x, w = lobatto(Q)
        assert abs(sum(w) - 2) < mp.mpf(10) ** (-40)
        out[f"lobatto_x_{Q}"], out[f"lobatto_w_{Q}"] = f64(x), f64(w)
    for P, Q in GAUSS_PAIRS:
        B, G = lagrange_tables(lobatto(P)[0], gauss(Q)[0])
        out[f"interp_gauss_{P}_{Q}"], out[f"grad_gauss_{P}_{Q}"] = f64(B), f64(G)
    for P, Q in CTOF_PAIRS:
        B, G = lagrange_tables(lobatto(P)[0], lobatto(Q)[0])
        out[f"interp_lobatto_{P}_{Q}"], out[f"grad_lobatto_{P}_{Q}"] = f64(B), f64(G)
    # cross-check against numpy's own Gauss rule before anything is written
    for Q in range(1, 9):
        xr, wr = np.polynomial.legendre.leggauss(Q)
        assert np.abs(out[f"gauss_x_{Q}"] - xr).max() < 2e-15 and np.abs(out[f"gauss_w_{Q}"] - wr).max() < 2e-15
    path = os.path.join(ROOT, "tests", "golden", "basis_tables.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
