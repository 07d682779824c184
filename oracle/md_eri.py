"""Independent McMurchie-Davidson integral engine (NumPy + SciPy Boys function).

TEST INFRASTRUCTURE ONLY.  It shares no code, no tables and no algorithm with the Rys-quadrature
path (oracle/jk_oracle.c, the HIP kernels): Hermite-Gaussian expansion + Boys function from
``scipy.special.hyp1f1``.  Used to (1) cross-check the Rys oracle for every angular momentum up to
g, (2) provide overlap / kinetic / nuclear-attraction matrices for the small RHF driver
(oracle/rhf.py) that reproduces the reference's hard-coded H2O energies.

Shell convention = the packed rows of ``BasisLayout`` (``[x,y,z,ao_loc,c0,e0,c1,e1,c2,e2,nprim,l]``),
Cartesian components in libcint order, s/p factors already folded into the coefficients.
"""
from functools import lru_cache
from math import pi

import numpy as np
from scipy.special import hyp1f1


def cart_powers(l):
    return [(lx, ly, l - lx - ly) for lx in range(l, -1, -1) for ly in range(l - lx, -1, -1)]


def boys(nmax, x):
    """F_0..F_nmax at x."""
    return np.array([hyp1f1(n + 0.5, n + 1.5, -x) / (2 * n + 1) for n in range(nmax + 1)])


def hermite_E(la, lb, a, b, xab):
    """E[i][j][t] for one Cartesian direction; xab = A - B."""
    p = a + b
    q = a * b / p
    E = np.zeros((la + 1, lb + 1, la + lb + 2))
    E[0, 0, 0] = np.exp(-q * xab * xab)
    xpa = -b / p * xab
    xpb = a / p * xab
    for i in range(la + 1):
        for j in range(lb + 1):
            if i == 0 and j == 0:
                continue
            for t in range(i + j + 1):
                if i > 0:
                    v = xpa * E[i - 1, j, t] + (t + 1) * E[i - 1, j, t + 1]
                    if t > 0:
                        v += E[i - 1, j, t - 1] / (2 * p)
                else:
                    v = xpb * E[i, j - 1, t] + (t + 1) * E[i, j - 1, t + 1]
                    if t > 0:
                        v += E[i, j - 1, t - 1] / (2 * p)
                E[i, j, t] = v
    return E


def hermite_R(lmax, alpha, rpc, scale_boys=None):
    """R[t,u,v] = R^0_{tuv}(alpha, rpc) for t+u+v <= lmax."""
    x = alpha * float(rpc @ rpc)
    F = boys(lmax, x) if scale_boys is None else scale_boys(lmax, x)
    Rn = np.zeros((lmax + 1, lmax + 1, lmax + 1, lmax + 1))
    for n in range(lmax + 1):
        Rn[n, 0, 0, 0] = (-2 * alpha) ** n * F[n]
    X, Y, Z = rpc
    for t in range(lmax + 1):
        for u in range(lmax + 1 - t):
            for v in range(lmax + 1 - t - u):
                if t + u + v == 0:
                    continue
                for n in range(lmax + 1 - (t + u + v)):
                    if t > 0:
                        val = X * Rn[n + 1, t - 1, u, v]
                        if t > 1:
                            val += (t - 1) * Rn[n + 1, t - 2, u, v]
                    elif u > 0:
                        val = Y * Rn[n + 1, t, u - 1, v]
                        if u > 1:
                            val += (u - 1) * Rn[n + 1, t, u - 2, v]
                    else:
                        val = Z * Rn[n + 1, t, u, v - 1]
                        if v > 1:
                            val += (v - 1) * Rn[n + 1, t, u, v - 2]
                    Rn[n, t, u, v] = val
    return Rn[0]


def _prims(row):
    n = int(row[10])
    return [(row[4 + 2 * p], row[5 + 2 * p]) for p in range(n)]


def _pair_E(la, lb, a, b, A, B):
    """Hermite coefficients of a primitive pair: Eab[ia, ib, t, u, v]."""
    Ex = hermite_E(la, lb, a, b, A[0] - B[0])
    Ey = hermite_E(la, lb, a, b, A[1] - B[1])
    Ez = hermite_E(la, lb, a, b, A[2] - B[2])
    pa, pb = cart_powers(la), cart_powers(lb)
    L = la + lb
    out = np.zeros((len(pa), len(pb), L + 1, L + 1, L + 1))
    for ia, (ax, ay, az) in enumerate(pa):
        for ib, (bx, by, bz) in enumerate(pb):
            out[ia, ib] = np.einsum("t,u,v->tuv", Ex[ax, bx, :L + 1], Ey[ay, by, :L + 1], Ez[az, bz, :L + 1])
    return out


def eri_block(rows, i, j, k, l, omega=0.0):
    """(ij|kl) Cartesian block from packed shell rows (contracted over <=3 primitives each)."""
    ri, rj, rk, rl = (np.asarray(rows[s], dtype=float) for s in (i, j, k, l))
    li, lj, lk, ll = (int(r[11]) for r in (ri, rj, rk, rl))
    A, B, C, D = ri[:3], rj[:3], rk[:3], rl[:3]
    nfi, nfj, nfk, nfl = (len(cart_powers(x)) for x in (li, lj, lk, ll))
    out = np.zeros((nfi, nfj, nfk, nfl))
    Lb, Lk = li + lj, lk + ll
    Lt = Lb + Lk
    sign = np.array([[[(-1.0) ** (t + u + v) for v in range(Lk + 1)] for u in range(Lk + 1)] for t in range(Lk + 1)])
    for ci, ai in _prims(ri):
        for cj, aj in _prims(rj):
            p = ai + aj
            P = (ai * A + aj * B) / p
            Eab = _pair_E(li, lj, ai, aj, A, B)
            for ck, ak in _prims(rk):
                for cl, al in _prims(rl):
                    q = ak + al
                    Q = (ak * C + al * D) / q
                    Ecd = _pair_E(lk, ll, ak, al, C, D) * sign
                    alpha = p * q / (p + q)
                    pref = 2 * pi ** 2.5 / (p * q * np.sqrt(p + q))
                    if omega and omega > 0:
                        # erf(omega r)/r: alpha -> alpha w^2/(w^2+alpha), prefactor sqrt of the same ratio
                        tf = omega * omega / (omega * omega + alpha)
                        R = hermite_R(Lt, alpha * tf, P - Q) * np.sqrt(tf)
                    else:
                        R = hermite_R(Lt, alpha, P - Q)
                    # W[t,u,v,T,U,V] = R[t+T,u+U,v+V]
                    idx_b = np.arange(Lb + 1)
                    idx_k = np.arange(Lk + 1)
                    W = R[np.ix_(range(Lt + 1), range(Lt + 1), range(Lt + 1))]
                    W = W[(idx_b[:, None, None, None, None, None] + idx_k[None, None, None, :, None, None]),
                          (idx_b[None, :, None, None, None, None] + idx_k[None, None, None, None, :, None]),
                          (idx_b[None, None, :, None, None, None] + idx_k[None, None, None, None, None, :])]
                    tmp = np.einsum("abtuv,tuvTUV->abTUV", Eab, W)
                    out += (ci * cj * ck * cl * pref) * np.einsum("abTUV,cdTUV->abcd", tmp, Ecd)
    return out


# ----------------------------------------------------------------------------- one-electron
def ovlp_kin_block(ra, rb):
    la, lb = int(ra[11]), int(rb[11])
    A, B = ra[:3], rb[:3]
    pa, pb = cart_powers(la), cart_powers(lb)
    S = np.zeros((len(pa), len(pb)))
    T = np.zeros_like(S)
    for ca, a in _prims(ra):
        for cb, b in _prims(rb):
            p = a + b
            E = [hermite_E(la, lb + 2, a, b, A[d] - B[d]) for d in range(3)]
            s1 = lambda d, i, j: E[d][i, j, 0] if j >= 0 else 0.0
            fac = ca * cb * (pi / p) ** 1.5

            def k1(d, i, j):
                v = -2 * b * b * s1(d, i, j + 2) + b * (2 * j + 1) * s1(d, i, j)
                if j >= 2:
                    v -= 0.5 * j * (j - 1) * s1(d, i, j - 2)
                return v
            for ia, pw_a in enumerate(pa):
                for ib, pw_b in enumerate(pb):
                    sx, sy, sz = (s1(d, pw_a[d], pw_b[d]) for d in range(3))
                    S[ia, ib] += fac * sx * sy * sz
                    T[ia, ib] += fac * (k1(0, pw_a[0], pw_b[0]) * sy * sz + sx * k1(1, pw_a[1], pw_b[1]) * sz
                                        + sx * sy * k1(2, pw_a[2], pw_b[2]))
    return S, T


def nuc_block(ra, rb, coords, charges):
    la, lb = int(ra[11]), int(rb[11])
    A, B = ra[:3], rb[:3]
    L = la + lb
    V = np.zeros((len(cart_powers(la)), len(cart_powers(lb))))
    for ca, a in _prims(ra):
        for cb, b in _prims(rb):
            p = a + b
            P = (a * A + b * B) / p
            Eab = _pair_E(la, lb, a, b, A, B)
            acc = np.zeros((L + 1, L + 1, L + 1))
            for C, Z in zip(coords, charges):
                acc -= Z * hermite_R(L, p, P - C)
            V += ca * cb * (2 * pi / p) * np.einsum("abtuv,tuv->ab", Eab, acc)
    return V


def int1e(rows, ao_loc, coords, charges):
    """S, T, V in the Cartesian AO space defined by rows/ao_loc (rows with zero width skipped)."""
    rows = np.asarray(rows, dtype=float)
    nao = int(ao_loc[-1])
    S = np.zeros((nao, nao))
    T = np.zeros((nao, nao))
    V = np.zeros((nao, nao))
    n = rows.shape[0]
    for i in range(n):
        i0, i1 = ao_loc[i], ao_loc[i + 1]
        if i1 == i0:
            continue
        for j in range(i + 1):
            j0, j1 = ao_loc[j], ao_loc[j + 1]
            if j1 == j0:
                continue
            s, t = ovlp_kin_block(rows[i], rows[j])
            v = nuc_block(rows[i], rows[j], coords, charges)
            S[i0:i1, j0:j1] = s
            S[j0:j1, i0:i1] = s.T
            T[i0:i1, j0:j1] = t
            T[j0:j1, i0:i1] = t.T
            V[i0:i1, j0:j1] = v
            V[j0:j1, i0:i1] = v.T
    return S, T, V
