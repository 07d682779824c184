"""Scalar ECP integrals on the device (role of ``/root/reference/jqc/backend/ecp.py``).

``get_ecp(mol_or_basis_layout, precision="fp64")`` has the reference's signature and meaning (``ecp.py:1371-1503``): the matrix
``<a| U_ECP |b>`` summed over every atom that carries a potential, in the molecule's own AO basis (``mol.intor("ECPscalar")``
order), as a device array.  Host side: tasks = (shell i <= shell j of the split / sorted layout, ECP atom) as in the reference's
``make_ecp_tasks`` (``ecp.py:1346-1369``; no screening there either), the potentials flattened per primitive from
``mol._ecpbas``, a radial Gauss-Chebyshev grid and the polynomial coefficients of the real spherical harmonics, both generated
here from closed forms; then ONE launch of ``ecp_scalar_kernel`` (``csrc/ecp_kernels.inc``: what replaces the reference's
per-(li, lj, l) JIT kernels and its tabulated angular coefficients).  Derivative integrals (``get_ecp_ip`` / ``get_ecp_ipip``,
``ecp.py:1506-1569``) are not built.

Parity: the reference compares against libcint's ``ECPscalar`` (third party, absent here) and stores no numbers, so this
row is checked against ``oracle/ecp.py`` (the definition by brute-force quadrature) only -- PARITY UNPINNED.
"""
import numpy as np

from . import lib as _lib

NR_DEFAULT = 160


def radial_grid(n=NR_DEFAULT):
    """(r, w): Gauss-Chebyshev points of the second kind on (-1, 1) mapped to (0, inf) by
    r = 2 ((1 + x) / 2)^3 ln(2 / (1 - x)) / ln 2 -- logarithmic towards infinity like the maps of Treutler and Ahlrichs, cubic at
    the origin so that an integrand that does NOT vanish at r = 0 (a 1 / r^2 term of a potential between s functions on the ECP
    centre) still meets the endpoint condition of the Chebyshev rule; w = dr, without the r^2 of the volume element.  With 160
    points int r^n exp(-a r^2) dr is exact to 4e-15 for a = 0.1 ... 5e4, n = 0 ... 8 (tests/test_ecp_oracle.py)."""
    i = np.arange(1, n + 1)
    x = np.cos(i * np.pi / (n + 1))
    wx = np.pi / (n + 1) * np.sin(i * np.pi / (n + 1))                  # weights of dx (Chebyshev weight sqrt(1 - x^2) divided out)
    ln2 = np.log(2.0)
    lg = np.log(2.0 / (1.0 - x))
    h = 0.5 * (1.0 + x)
    r = 2.0 * h ** 3 / ln2 * lg
    dr = 2.0 * (1.5 * h * h * lg + h ** 3 / (1.0 - x)) / ln2
    return r[::-1].copy(), (wx * dr)[::-1].copy()


def ylm_table(lmax=4):
    """[(lmax + 1)^2, 15]: coefficients of the degree-l Cartesian monomials (libcint order) in the ORTHONORMAL real spherical
    harmonics on the unit sphere, from the closed form of gto/c2s.py (l = 0, 1 carry their factor explicitly there)."""
    from ..gto import c2s
    out = np.zeros(((lmax + 1) ** 2, 15))
    for l in range(lmax + 1):
        C = c2s.cart2sph_l(l) * c2s.fac_sp(l)            # [ncart, 2l + 1]
        out[l * l:(l + 1) * (l + 1), :C.shape[0]] = C.T
    return out


def ecp_arrays(mol):
    """(xyz [natm_ecp, 3], loc [natm_ecp + 1], terms [nterm, 4] = {l, power, zeta, coef}) from ``mol._ecpbas``."""
    from ..gto import ecp as gecp
    ch = gecp.channels(mol)
    xyz, loc, terms = [], [0], []
    coords = np.asarray(mol.atom_coords(), dtype=float)
    for ia in sorted(ch):
        xyz.append(coords[ia])
        for l, power, zeta, coef in ch[ia]:
            assert -1 <= l <= 4, "ECP projectors up to l = 4"
            terms += [[float(l), float(power), float(z), float(c)] for z, c in zip(zeta, coef)]
        loc.append(len(terms))
    return np.asarray(xyz, dtype=float).reshape(-1, 3), np.asarray(loc, dtype=np.int32), np.asarray(terms, dtype=float).reshape(-1, 4)


def get_ecp(mol_or_basis_layout, precision="fp64", nr=NR_DEFAULT):
    import torch
    if precision != "fp64":
        raise ValueError("Only double precision ('fp64') is supported for ECP kernels")      # (reference jqc/pyscf/ecp.py:51-53)
    if hasattr(mol_or_basis_layout, "packed"):
        layout = mol_or_basis_layout
        mol = layout._mol
    else:
        from ..pyscf.basis import BasisLayout
        mol = mol_or_basis_layout
        layout = BasisLayout.from_mol(mol, alignment=1)
    dev = _lib.require_gpu()
    nao_mol = mol.nao
    if getattr(mol, "_ecpbas", None) is None or len(mol._ecpbas) == 0:
        return torch.zeros((nao_mol, nao_mol), dtype=torch.float64, device=dev)
    xyz, loc, terms = ecp_arrays(mol)
    shells = np.nonzero(~np.asarray(layout.pad_id))[0]
    assert int(np.max(np.asarray(layout.angs)[shells])) <= 4, "ECP kernels: shells up to l = 4"
    i, j = np.triu_indices(len(shells))
    pairs = np.stack([shells[i], shells[j]], 1)
    tasks = np.concatenate([np.concatenate([pairs, np.full((len(pairs), 1), k)], 1) for k in range(len(xyz))]).astype(np.int32)
    r, w = radial_grid(nr)
    nao = int(layout.nao)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mat = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    keep = [t(tasks), t(xyz), t(loc), t(terms), t(r), t(w), t(ylm_table())]
    b64 = layout.basis_data_fp64["packed"]
    _lib.check(_lib.lib().jqc_ecp_scalar(b64.data_ptr(), nao, keep[0].data_ptr(), int(tasks.shape[0]), keep[1].data_ptr(),
                                         keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), int(nr),
                                         keep[6].data_ptr(), mat.data_ptr(), _lib.stream_ptr()))
    out = layout.dm_to_mol(mat.reshape(1, nao, nao))[0]
    torch.cuda.current_stream().synchronize()          # (the temporaries above must outlive the launch)
    return out
