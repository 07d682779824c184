#!/usr/bin/env python3
"""Generate the Rys-quadrature Chebyshev tables used by the HIP kernels and the CPU oracle.

This is an independent derivation (mpmath, 80 digits), NOT a copy of the reference's
``jqc/backend/rys/rys_root{n}.cu`` tables.  What is reproduced from the reference is only the
*interface* of its ``rys_roots`` routine (``/root/reference/jqc/backend/rys/rys_roots.cu:30-160``):

  * roots are returned as t^2 and weights as w, so that  sum_i w_i f(t_i^2) = int_0^1 exp(-x t^2) f(t^2) dt
  * piecewise Chebyshev of degree 13 on intervals of width 2.5 in x, up to x = 5*nroots + 35
  * beyond that the asymptotic (half Gauss-Hermite) form  t_i^2 = r_i / x,  w_i = v_i / sqrt(x)

Method: moments m_k = F_k(x) (Boys function) -> Hankel matrix -> Cholesky -> Jacobi matrix
(Golub-Welsch) -> eigen-decomposition, all at 80 significant digits.  Chebyshev coefficients by
discrete orthogonality on 32 Chebyshev nodes per interval, truncated to 14 terms.

Output file (little-endian):  joltqc_amd/data/rys_tables.npz with
    cheb_{n}   float64[nint(n), n, 14, 2]   (interval, root, coefficient k, {root, weight});
                                            value = c_0 + sum_{k>=1} c_k T_k(u),  u = (x - 2.5 it)*0.8 - 1
    large_{n}  float64[n, 2]                ({r_i, v_i}) for the asymptotic branch
and a flat packed blob (see pack_tables in joltqc_amd/backend/rys.py) is derived from it at import.
"""
import sys
import numpy as np
from multiprocessing import Pool
from mpmath import mp

DEGREE = 13
NCOEF = DEGREE + 1
WIDTH = 2.5
NNODE = 32
NMAX = 9
mp.dps = 80


def nintervals(n):
    # covers [0, 5n+35]
    return 2 * n + 14


def rys_mp(n, x):
    """Rys roots (as t^2) and weights for exp(-x t^2) on t in [0,1]; mpmath numbers."""
    x = mp.mpf(x)
    half = mp.mpf(1) / 2
    if x == 0:
        m = [mp.mpf(1) / (2 * k + 1) for k in range(2 * n + 1)]
    else:
        m = [mp.gammainc(k + half, 0, x) / (2 * x ** (k + half)) for k in range(2 * n + 1)]
    H = mp.matrix(n + 1, n + 1)
    for i in range(n + 1):
        for j in range(n + 1):
            H[i, j] = m[i + j]
    R = mp.cholesky(H).T
    J = mp.matrix(n, n)
    for j in range(n):
        a = R[j, j + 1] / R[j, j]
        if j > 0:
            a -= R[j - 1, j] / R[j - 1, j - 1]
        J[j, j] = a
        if j < n - 1:
            b = R[j + 1, j + 1] / R[j, j]
            J[j, j + 1] = b
            J[j + 1, j] = b
    E, Q = mp.eigsy(J)
    order = sorted(range(n), key=lambda i: E[i])
    roots = [E[i] for i in order]
    weights = [m[0] * Q[0, i] ** 2 for i in order]
    return roots, weights


def _job(args):
    n, it = args
    mp.dps = 80
    # Chebyshev nodes of the first kind
    ks = [mp.cos(mp.pi * (mp.mpf(j) + mp.mpf(1) / 2) / NNODE) for j in range(NNODE)]
    vals_r = []
    vals_w = []
    for u in ks:
        x = (u + 1) * (WIDTH / 2) + WIDTH * it
        r, w = rys_mp(n, x)
        vals_r.append(r)
        vals_w.append(w)
    out = np.zeros((n, NCOEF, 2))
    for i in range(n):
        for k in range(NCOEF):
            cr = mp.mpf(0)
            cw = mp.mpf(0)
            for j in range(NNODE):
                tk = mp.cos(k * mp.pi * (mp.mpf(j) + mp.mpf(1) / 2) / NNODE)
                cr += vals_r[j][i] * tk
                cw += vals_w[j][i] * tk
            fac = mp.mpf(1 if k == 0 else 2) / NNODE
            out[i, k, 0] = float(cr * fac)
            out[i, k, 1] = float(cw * fac)
    return n, it, out


def large_x(n):
    """positive half of the 2n-point Gauss-Hermite rule: r_i = y_i^2, v_i = w_i."""
    mp.dps = 80
    N = 2 * n
    J = mp.matrix(N, N)
    for j in range(N - 1):
        b = mp.sqrt(mp.mpf(j + 1) / 2)
        J[j, j + 1] = b
        J[j + 1, j] = b
    E, Q = mp.eigsy(J)
    m0 = mp.sqrt(mp.pi)
    pairs = sorted((E[i], m0 * Q[0, i] ** 2) for i in range(N) if E[i] > 0)
    return np.array([[float(y * y), float(w)] for y, w in pairs])


def main(out):
    jobs = [(n, it) for n in range(1, NMAX + 1) for it in range(nintervals(n))]
    tabs = {n: np.zeros((nintervals(n), n, NCOEF, 2)) for n in range(1, NMAX + 1)}
    with Pool(8) as pool:
        for n, it, arr in pool.imap_unordered(_job, jobs, chunksize=2):
            tabs[n][it] = arr
    data = {}
    for n in range(1, NMAX + 1):
        data[f"cheb_{n}"] = tabs[n]
        data[f"large_{n}"] = large_x(n)
    np.savez(out, **data)
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "joltqc_amd/data/rys_tables.npz")
