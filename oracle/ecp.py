"""Brute-force CPU oracle of the scalar ECP integrals (NumPy).  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

What is computed (definition, as in the reference's path ``get_ecp`` -> ``ECPscalar``, /root/reference/jqc/backend/ecp.py:1371-1503;
the kernels ``ecp/ecp_type1.cu`` / ``ecp_type2.cu`` evaluate it semi-analytically): for every atom C that carries a potential

    V_ab = <a| U_L(r_C) |b>  +  sum_l <a| U_l(r_C) sum_m |l m><l m| |b>,        U(r) = sum_k c_k r^(n_k - 2) exp(-zeta_k r^2)

with the local channel U_L (``ANG_OF = -1``) acting on everything and the semi-local channels through the projectors onto the
spherical harmonics around C.  The reference's own tests compare against libcint's ``mol.intor("ECPscalar")``
(``jqc/pyscf/tests/test_ecp_small.py:118-131``); libcint and PySCF are third party and absent from this image, and the tests
hold no stored integrals (one total energy depends on them: I2 / def2-TZVPP + def2 ECP, PBE, -582.7625143308, test_dft_ecp.py:52-56 -- iodine's
basis and ECP tables are not in this image), so NO reference-held value pins this row: the oracle is the definition evaluated by plain quadrature,
checked against closed forms where they exist (tests/test_ecp_oracle.py).

Method (deliberately different from the device kernels: no Bessel functions, no angular tables, no binomial expansions):
a product grid around C -- Gauss-Legendre panels in r, Gauss-Legendre in cos(theta), uniform in phi -- on which the AOs are
evaluated directly (oracle/dft.py).  Local channel: sum_g w_g U_L(r_g) phi_a phi_b.  Semi-local: per radius the projections
A_a,lm(r) = sum_Omega w_Omega Y_lm(Omega) phi_a(C + r Omega), then sum_r w_r r^2 U_l(r) A_a,lm A_b,lm.  The angular order is
raised until the result is stable (``ecp_scalar(..., nang=...)``); the cost is O(nao x points), fine for the test molecules.
"""
import numpy as np

from . import dft


def _radial(nodes_per_panel=48, edges=(0.0, 0.05, 0.2, 0.5, 1.0, 2.0, 3.5, 6.0, 10.0, 16.0)):
    x, w = np.polynomial.legendre.leggauss(nodes_per_panel)
    rs, ws = [], []
    for a, b in zip(edges[:-1], edges[1:]):
        rs.append(0.5 * (b - a) * x + 0.5 * (a + b))
        ws.append(0.5 * (b - a) * w)
    return np.concatenate(rs), np.concatenate(ws)


def _angular(ntheta):
    ct, wt = np.polynomial.legendre.leggauss(ntheta)
    nphi = 2 * ntheta
    phi = (np.arange(nphi) + 0.5) * 2 * np.pi / nphi
    st = np.sqrt(1 - ct * ct)
    xyz = np.stack([np.outer(st, np.cos(phi)).ravel(), np.outer(st, np.sin(phi)).ravel(), np.repeat(ct, nphi)], 1)
    return xyz, np.repeat(wt, nphi) * (2 * np.pi / nphi)


def real_sph_harm(l, xyz):
    """Orthonormal real spherical harmonics of degree l on unit vectors xyz [n, 3]: [2l+1, n] (any orthonormal basis of the
    degree-l subspace gives the same projector; this one comes from scipy's complex Y_l^m)."""
    import scipy.special as sp
    theta = np.arccos(np.clip(xyz[:, 2], -1, 1))
    phi = np.arctan2(xyz[:, 1], xyz[:, 0])
    # (scipy >= 1.15: sph_harm_y(l, m, polar, azimuth); older: sph_harm(m, l, azimuth, polar))
    ycomplex = (lambda m: sp.sph_harm_y(l, m, theta, phi)) if hasattr(sp, "sph_harm_y") else (lambda m: sp.sph_harm(m, l, phi, theta))
    out = [np.real(ycomplex(0))]
    for m in range(1, l + 1):
        y = ycomplex(m)
        out.append(np.sqrt(2.0) * np.real(y))
        out.append(np.sqrt(2.0) * np.imag(y))
    return np.asarray(out)


def _u(r, terms):
    """U(r) of one channel: terms = [(power, zeta[], coef[]), ...]."""
    v = np.zeros_like(r)
    for power, zeta, coef in terms:
        for z, c in zip(zeta, coef):
            v += c * r ** (power - 2) * np.exp(-z * r * r)
    return v


def ecp_scalar(layout, channels, coords_of_atom, nang=64, nrad=48):
    """ECP matrix in the INTERNAL Cartesian AO order of ``layout`` (joltqc_amd.pyscf.basis.BasisLayout): [nao_int, nao_int].
    ``channels`` = joltqc_amd.gto.ecp.channels(mol); ``coords_of_atom[ia]`` = centre of atom ia (Bohr)."""
    nao = int(layout.ao_loc[-1])
    V = np.zeros((nao, nao))
    r, wr = _radial(nrad)
    ang, wang = _angular(nang)
    for ia, rows in channels.items():
        C = np.asarray(coords_of_atom[ia], dtype=float)
        by_l = {}
        for l, power, zeta, coef in rows:
            by_l.setdefault(l, []).append((power, zeta, coef))
        ylm = {l: real_sph_harm(l, ang) for l in by_l if l >= 0}
        ul = {l: _u(r, t) for l, t in by_l.items()}
        for n in range(len(r)):
            pts = C + r[n] * ang
            ao = dft.eval_ao_cart(layout.packed, layout.ao_loc, pts)[0]        # [nao, nang_pts]
            if -1 in by_l:
                V += (wr[n] * r[n] ** 2 * ul[-1][n]) * (ao * wang) @ ao.T
            for l, y in ylm.items():
                A = (ao * wang) @ y.T                                           # [nao, 2l+1]
                V += (wr[n] * r[n] ** 2 * ul[l][n]) * A @ A.T
    return V


def ecp_scalar_mol(layout, mol, nang=64, nrad=48):
    """The same matrix in the molecule's own AO basis (spherical or Cartesian), as ``mol.intor("ECPscalar")`` lays it out."""
    from joltqc_amd.gto import ecp as gecp
    V = ecp_scalar(layout, gecp.channels(mol), mol.atom_coords(), nang, nrad)
    T = layout.transform_matrix()
    return T.T @ V @ T


def ecp_ip(layout, channels, coords_of_atom, atom, nang=64, nrad=48):
    """<d/dr a| U_C |b> for ONE ECP atom C = ``atom``: [3, nao_int, nao_int] in the internal Cartesian AO order (what libcint's
    ``ECPscalar_iprinv`` is for the nucleus selected by ``with_rinv_at_nucleus``; reference get_ecp_ip, backend/ecp.py:953-1138).
    Same quadrature as ``ecp_scalar`` with the AO gradients of oracle/dft.py on the bra side."""
    nao = int(layout.ao_loc[-1])
    V = np.zeros((3, nao, nao))
    r, wr = _radial(nrad)
    ang, wang = _angular(nang)
    C = np.asarray(coords_of_atom[atom], dtype=float)
    by_l = {}
    for l, power, zeta, coef in channels[atom]:
        by_l.setdefault(l, []).append((power, zeta, coef))
    ylm = {l: real_sph_harm(l, ang) for l in by_l if l >= 0}
    ul = {l: _u(r, t) for l, t in by_l.items()}
    for n in range(len(r)):
        ao = dft.eval_ao_cart(layout.packed, layout.ao_loc, C + r[n] * ang, deriv=1)        # [4, nao, npts]
        for x in range(3):
            if -1 in by_l:
                V[x] += (wr[n] * r[n] ** 2 * ul[-1][n]) * (ao[1 + x] * wang) @ ao[0].T
            for l, y in ylm.items():
                V[x] += (wr[n] * r[n] ** 2 * ul[l][n]) * ((ao[1 + x] * wang) @ y.T) @ ((ao[0] * wang) @ y.T).T
    return V


def ecp_ip_mol(layout, mol, atom, nang=64, nrad=48):
    from joltqc_amd.gto import ecp as gecp
    V = ecp_ip(layout, gecp.channels(mol), mol.atom_coords(), atom, nang, nrad)
    T = layout.transform_matrix()
    return np.einsum("pi,xpq,qj->xij", T, V, T)


def ao_second_derivatives(packed, ao_loc, coords):
    """d^2 phi / dx_i dx_j of the internal Cartesian AOs on ``coords``: [3, 3, nao_int, npts].  Each AO is a sum over primitives of
    products of 1-D factors g(x) = x^n exp(-a x^2), differentiated factor by factor (checked against central differences of
    oracle/dft.py's first derivatives in tests/test_ecp_oracle.py)."""
    packed = np.asarray(packed, dtype=float)
    coords = np.asarray(coords, dtype=float)
    nao, ng = int(ao_loc[-1]), coords.shape[0]
    out = np.zeros((3, 3, nao, ng))

    def g(x, n, a, order):
        e = np.exp(-a * x * x)
        pw = lambda k: x ** k if k > 0 else (np.ones_like(x) if k == 0 else np.zeros_like(x))
        if order == 0:
            return pw(n) * e
        if order == 1:
            return (n * pw(n - 1) - 2 * a * pw(n + 1)) * e
        return (n * (n - 1) * pw(n - 2) - 2 * a * (2 * n + 1) * pw(n) + 4 * a * a * pw(n + 2)) * e

    for s in range(packed.shape[0]):
        if ao_loc[s + 1] == ao_loc[s]:
            continue
        l, npr = int(packed[s, 11]), int(packed[s, 10])
        r = coords - packed[s, :3]
        for n, pows in enumerate(dft.cart_powers(l)):
            for p in range(npr):
                c, a = packed[s, 4 + 2 * p], packed[s, 5 + 2 * p]
                f = [[g(r[:, d], pows[d], a, o) for o in range(3)] for d in range(3)]
                for i in range(3):
                    for j in range(3):
                        order = [(d == i) + (d == j) for d in range(3)]
                        out[i, j, ao_loc[s] + n] += c * f[0][order[0]] * f[1][order[1]] * f[2][order[2]]
    return out


def ecp_ipip(layout, channels, coords_of_atom, atom, ip_type="ipipv", nang=64, nrad=48):
    """Second derivatives for ONE ECP atom, [3, 3, nao_int, nao_int] (internal Cartesian AO order): "ipipv" = <d_i d_j a| U_C |b>,
    "ipvip" = <d_i a| U_C |d_j b> (libcint's ``ECPscalar_ipiprinv`` / ``ECPscalar_iprinvip`` for the nucleus selected by
    ``with_rinv_at_nucleus``; reference get_ecp_ipip, backend/ecp.py:1141-1340).  Same quadrature as ``ecp_scalar``."""
    assert ip_type in ("ipipv", "ipvip")
    nao = int(layout.ao_loc[-1])
    V = np.zeros((3, 3, nao, nao))
    r, wr = _radial(nrad)
    ang, wang = _angular(nang)
    C = np.asarray(coords_of_atom[atom], dtype=float)
    by_l = {}
    for l, power, zeta, coef in channels[atom]:
        by_l.setdefault(l, []).append((power, zeta, coef))
    ylm = {l: real_sph_harm(l, ang) for l in by_l if l >= 0}
    ul = {l: _u(r, t) for l, t in by_l.items()}
    for n in range(len(r)):
        pts = C + r[n] * ang
        ao = dft.eval_ao_cart(layout.packed, layout.ao_loc, pts, deriv=1)
        hess = ao_second_derivatives(layout.packed, layout.ao_loc, pts) if ip_type == "ipipv" else None
        for i in range(3):
            for j in range(3):
                bra, ket = (hess[i, j], ao[0]) if ip_type == "ipipv" else (ao[1 + i], ao[1 + j])
                if -1 in by_l:
                    V[i, j] += (wr[n] * r[n] ** 2 * ul[-1][n]) * (bra * wang) @ ket.T
                for l, y in ylm.items():
                    V[i, j] += (wr[n] * r[n] ** 2 * ul[l][n]) * ((bra * wang) @ y.T) @ ((ket * wang) @ y.T).T
    return V


def ecp_ipip_mol(layout, mol, atom, ip_type="ipipv", nang=64, nrad=48):
    """[9, nao, nao] in the molecule's own AO basis, component 3 i + j (the layout of ``mol.intor("ECPscalar_ipiprinv")``)."""
    from joltqc_amd.gto import ecp as gecp
    V = ecp_ipip(layout, gecp.channels(mol), mol.atom_coords(), atom, ip_type, nang, nrad)
    T = layout.transform_matrix()
    return np.einsum("pi,xypq,qj->xyij", T, V, T).reshape(9, T.shape[1], T.shape[1])
