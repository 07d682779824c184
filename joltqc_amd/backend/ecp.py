"""Scalar ECP integrals on the device (role of ``/root/reference/jqc/backend/ecp.py``).

``get_ecp(mol_or_basis_layout, precision="fp64")`` has the reference's signature and meaning (``ecp.py:1371-1503``): the matrix
``<a| U_ECP |b>`` summed over every atom that carries a potential, in the molecule's own AO basis (``mol.intor("ECPscalar")``
order), as a device array.  Host side: tasks = (shell i <= shell j of the split / sorted layout, ECP atom) as in the reference's
``make_ecp_tasks`` (``ecp.py:1346-1369``; no screening there either), the potentials flattened per primitive from
``mol._ecpbas``, a radial Gauss-Chebyshev grid and the polynomial coefficients of the real spherical harmonics, both generated
here from closed forms; then ONE launch of ``ecp_scalar_kernel`` (``csrc/ecp_kernels.inc``: what replaces the reference's
per-(li, lj, l) JIT kernels and its tabulated angular coefficients).  First derivatives: ``get_ecp_ip`` (``ecp.py:953-1138``: <grad a| U_C |b> per ECP atom C) through the SAME kernel -- the
gradient of a Cartesian Gaussian shell is an (l + 1) shell with coefficients -2 alpha_p c_p plus an (l - 1) shell with the
shell's own coefficients, so the bra shells are replaced by those auxiliary shells and the blocks are recombined on the host
(the reference has separate ``ecp_type{1,2}_ip.cu`` kernels for Cartesian molecules and computes spherical ones on the CPU,
``:985-1012``).  Second derivatives ``get_ecp_ipip`` (``ecp.py:1141-1340``: <grad grad a| U_C |b>
("ipipv") and <grad a| U_C |grad b> ("ipvip")) the same way with (l + 2, alpha^2 c), (l, alpha c), (l - 2, c) auxiliary shells, or l +- 1
shells on both sides.

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


SCREEN_EXPONENT = 46.0      # tasks whose integrand is below exp(-46) = 1e-20 of its prefactors everywhere are skipped


def _screened_pairs(rows_a, parents_a, rows_b, parents_b, packed, amin, centre, z, k):
    """(row a, row b, ECP atom k) triples that pass the distance bound of ``screen_tasks`` -- rows = shell-table rows (auxiliary shells carry
    their parent's centre and most diffuse exponent: ``parents_*``), vectorised: int32 [n, 3]."""
    da = np.linalg.norm(packed[parents_a, :3] - centre, axis=1)
    db = np.linalg.norm(packed[parents_b, :3] - centre, axis=1)
    aa, ab = amin[parents_a], amin[parents_b]
    bound = ((aa * da * da)[:, None] + (ab * db * db)[None, :]
             - np.add.outer(aa * da, ab * db) ** 2 / (np.add.outer(aa, ab) + z))
    ia, ib = np.nonzero(bound <= SCREEN_EXPONENT)
    return np.stack([rows_a[ia], rows_b[ib], np.full(ia.size, k)], 1).astype(np.int32).reshape(-1, 3)


def screen_tasks(layout, shells, xyz, terms, loc):
    """(shell i <= shell j, ECP atom k) triples worth evaluating.  The reference's make_ecp_tasks keeps every triple ("TODO: Add
    screening here", ecp.py:1355); the potential is short-ranged, so for a large molecule almost all of them are empty.  Bound: with
    the most diffuse exponents a, b of the two shells at distances d_a, d_b from the ECP atom and its most diffuse exponent z, the
    radial integrand carries exp(-a (r - d_a)^2 - b (r - d_b)^2 - z r^2) (the Bessel functions are bounded by exp(kappa), which
    is what turns exp(-a (r^2 + d_a^2)) into the shifted Gaussian); its exponent is smallest at r* = (a d_a + b d_b) / (a + b + z).
    A task is dropped when that minimum exceeds SCREEN_EXPONENT."""
    packed = np.asarray(layout.packed)
    amin = np.array([packed[s, 5:5 + 2 * int(packed[s, 10]):2].min() for s in shells])
    pos = packed[shells, :3]
    i, j = np.triu_indices(len(shells))
    out = []
    for k in range(len(xyz)):
        z = float(terms[loc[k]:loc[k + 1], 2].min())
        d = np.linalg.norm(pos - xyz[k], axis=1)
        a, b, da, db = amin[i], amin[j], d[i], d[j]
        fmin = a * da * da + b * db * db - (a * da + b * db) ** 2 / (a + b + z)
        keep = fmin <= SCREEN_EXPONENT
        out.append(np.stack([shells[i[keep]], shells[j[keep]], np.full(int(keep.sum()), k)], 1))
    return np.concatenate(out).astype(np.int32) if out else np.zeros((0, 3), np.int32)


def get_ecp(mol_or_basis_layout, precision="fp64", nr=NR_DEFAULT, screen=True):
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
    if screen:
        tasks = screen_tasks(layout, shells, xyz, terms, loc)
    else:
        i, j = np.triu_indices(len(shells))
        pairs = np.stack([shells[i], shells[j]], 1)
        tasks = np.concatenate([np.concatenate([pairs, np.full((len(pairs), 1), k)], 1) for k in range(len(xyz))]).astype(np.int32)
    get_ecp.last_ntasks = int(tasks.shape[0])
    r, w = radial_grid(nr)
    nao = int(layout.nao)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mat = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    keep = [t(tasks), t(xyz), t(loc), t(terms), t(r), t(w), t(ylm_table())]
    b64 = layout.basis_data_fp64["packed"]
    _lib.check(_lib.lib().jqc_ecp_scalar(b64.data_ptr(), nao, keep[0].data_ptr(), int(tasks.shape[0]), keep[1].data_ptr(),
                                         keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), int(nr),
                                         keep[6].data_ptr(), mat.data_ptr(), 1, 4, _lib.stream_ptr()))
    out = layout.dm_to_mol(mat.reshape(1, nao, nao))[0]
    torch.cuda.current_stream().synchronize()          # (the temporaries above must outlive the launch)
    return out


def _cart(l):
    return [(lx, ly, l - lx - ly) for lx in range(l, -1, -1) for ly in range(l - lx, -1, -1)]


def get_ecp_ip(mol_or_basis_layout, ip_type="ip", ecp_atoms=None, precision="fp64", nr=NR_DEFAULT):
    """``[n_ecp_atoms, 3, nao, nao]``: <d/dr a| U_C |b> for every ECP atom C (reference ``get_ecp_ip``, the per-atom blocks its tests
    compare with libcint's ``ECPscalar_iprinv`` under ``with_rinv_at_nucleus(C)``, jqc/pyscf/tests/test_ecp_small.py:133-147), in the
    molecule's own AO basis, spherical or Cartesian."""
    import torch
    if ip_type != "ip":
        raise ValueError(f"Invalid ip_type: {ip_type}. Only 'ip' is supported.")
    if precision != "fp64":
        raise ValueError("Only double precision ('fp64') is supported for ECP kernels")
    if hasattr(mol_or_basis_layout, "packed"):
        layout, mol = mol_or_basis_layout, mol_or_basis_layout._mol
    else:
        from ..pyscf.basis import BasisLayout
        mol = mol_or_basis_layout
        layout = BasisLayout.from_mol(mol, alignment=1)
    dev = _lib.require_gpu()
    nao_mol = mol.nao
    have = getattr(mol, "_ecpbas", None) is not None and len(mol._ecpbas) > 0
    all_atoms = sorted({int(a) for a in np.asarray(mol._ecpbas)[:, 0]}) if have else []
    want = all_atoms if ecp_atoms is None else [a for a in ecp_atoms if a in all_atoms]
    if not want:
        return torch.zeros((0 if ecp_atoms is None else len(ecp_atoms), 3, nao_mol, nao_mol), dtype=torch.float64, device=dev)
    xyz, loc, terms = ecp_arrays(mol)
    packed = np.asarray(layout.packed)
    shells = np.nonzero(~np.asarray(layout.pad_id))[0]
    assert int(np.max(np.asarray(layout.angs)[shells])) <= 4, "ECP kernels: shells up to l = 4"
    nao = int(layout.nao)
    # auxiliary bra shells: (l + 1) with coefficients alpha_p c_p (the -2 goes into the recombination), (l - 1) with c_p
    rows, off = [packed[s].copy() for s in range(packed.shape[0])], nao
    plus_of, minus_of = {}, {}
    for s in shells:
        l, npr = int(packed[s, 11]), int(packed[s, 10])
        for dl, store in ((1, plus_of), (-1, minus_of)):
            if l + dl < 0:
                continue
            r = packed[s].copy()
            r[11] = l + dl
            r[3] = off
            if dl == 1:
                r[4:4 + 2 * npr:2] = packed[s, 4:4 + 2 * npr:2] * packed[s, 5:5 + 2 * npr:2]
            store[int(s)] = (len(rows), off)
            rows.append(r)
            off += (l + dl + 1) * (l + dl + 2) // 2
    ntot = off
    table = np.ascontiguousarray(np.stack(rows))
    # recombination: row of d/dx_dir phi_(s, comp) = n_dir phi-_(comp - e_dir) - 2 phi+_(comp + e_dir)
    ip_, im_ = np.zeros((3, nao), dtype=np.int64), np.full((3, nao), ntot, dtype=np.int64)      # (ntot: a row of zeros)
    wm = np.zeros((3, nao))
    for s in shells:
        l, a0 = int(packed[s, 11]), int(packed[s, 3])
        cp_, cm_ = {c: n for n, c in enumerate(_cart(l + 1))}, ({c: n for n, c in enumerate(_cart(l - 1))} if l > 0 else {})
        for n, c in enumerate(_cart(l)):
            for d in range(3):
                up = tuple(c[x] + (x == d) for x in range(3))
                ip_[d, a0 + n] = plus_of[int(s)][1] + cp_[up]
                if c[d] > 0:
                    dn = tuple(c[x] - (x == d) for x in range(3))
                    im_[d, a0 + n] = minus_of[int(s)][1] + cm_[dn]
                    wm[d, a0 + n] = c[d]
    aux = [idx for s in shells for idx in ([plus_of[int(s)][0]] + ([minus_of[int(s)][0]] if int(s) in minus_of else []))]
    # the distance screening of the value integrals, shell by shell (an auxiliary shell has its parent's centre and exponents)
    parent = {plus_of[int(s)][0]: int(s) for s in shells}
    parent.update({minus_of[int(s)][0]: int(s) for s in shells if int(s) in minus_of})
    amin_arr = np.zeros(packed.shape[0])
    for s in shells:
        amin_arr[s] = packed[s, 5:5 + 2 * int(packed[s, 10]):2].min()
    r, w = radial_grid(nr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tab_d, loc_d, terms_d, xyz_d, r_d, w_d, ylm_d = t(table), t(loc), t(terms), t(xyz), t(r), t(w), t(ylm_table())
    ip_d, im_d, wm_d = t(ip_), t(im_), t(wm)
    ecp_index = {a: k for k, a in enumerate(all_atoms)}
    out = torch.zeros((len(want), 3, nao_mol, nao_mol), dtype=torch.float64, device=dev)
    for n, atom in enumerate(want):
        k = ecp_index[atom]
        z = float(terms[loc[k]:loc[k + 1], 2].min())
        # (vectorised like screen_tasks: one distance vector per side, the bound as an outer sum -- advisor finding of round 5: the
        #  Python double loop cost tens of seconds per ECP atom at 100 atoms)
        tasks = _screened_pairs(np.asarray(aux), np.asarray([parent[ia] for ia in aux]), np.asarray(shells), np.asarray(shells),
                                packed, amin_arr, xyz[k], z, k)
        mat = torch.zeros((ntot + 1, ntot), dtype=torch.float64, device=dev)
        if len(tasks):
            tk = t(np.asarray(tasks, dtype=np.int32))
            _lib.check(_lib.lib().jqc_ecp_scalar(tab_d.data_ptr(), ntot, tk.data_ptr(), int(tasks.shape[0]), xyz_d.data_ptr(), loc_d.data_ptr(),
                                                 terms_d.data_ptr(), r_d.data_ptr(), w_d.data_ptr(), int(nr), ylm_d.data_ptr(),
                                                 mat.data_ptr(), 0, 5, _lib.stream_ptr()))
        cart = torch.stack([wm_d[d][:, None] * mat[im_d[d], :nao] - 2.0 * mat[ip_d[d], :nao] for d in range(3)])
        out[n] = layout.dm_to_mol(cart)
        torch.cuda.current_stream().synchronize()
    return out


def ecp_energy_per_atom(mol_or_basis_layout, dm, nr=NR_DEFAULT):
    """``[natm, 3]``: d/dR tr(D h_ECP) at fixed density matrix (the ECP term of a nuclear gradient, in the form the two-electron
    term is offered by ``joltqc_amd.pyscf.grad``; the reference leaves forces to GPU4PySCF, which builds this from the same
    ``ECPscalar_iprinv`` blocks, cf. jqc/backend/ecp.py:953-968).  With ip_C = <grad a| U_C |b> (``get_ecp_ip``) and a symmetric D:
    moving the shells of atom R contributes -2 sum_{a on R} sum_b D_ab ip_C[a, b] for every ECP atom C, and moving the potential of
    C itself, by translational invariance, +2 sum_ab D_ab ip_C[a, b]."""
    import torch
    if hasattr(mol_or_basis_layout, "packed"):
        layout, mol = mol_or_basis_layout, mol_or_basis_layout._mol
    else:
        from ..pyscf.basis import BasisLayout
        mol = mol_or_basis_layout
        layout = BasisLayout.from_mol(mol, alignment=1)
    dev = _lib.require_gpu()
    natm = int(mol.natm)
    out = torch.zeros((natm, 3), dtype=torch.float64, device=dev)
    if getattr(mol, "_ecpbas", None) is None or len(mol._ecpbas) == 0:
        return out
    d = torch.as_tensor(np.asarray(dm) if not torch.is_tensor(dm) else dm, dtype=torch.float64, device=dev)
    d = 0.5 * (d + d.T)
    ip = get_ecp_ip(layout, nr=nr)                                          # [n_ecp, 3, nao, nao]
    ecp_atoms = sorted({int(a) for a in np.asarray(mol._ecpbas)[:, 0]})
    # atom of every AO of the molecule's own basis
    bas = np.asarray(mol._bas)
    loc = np.asarray(mol.ao_loc_nr())
    ao_atom = np.concatenate([np.full(int(loc[i + 1] - loc[i]), int(bas[i, 0])) for i in range(bas.shape[0])])
    onehot = torch.zeros((natm, d.shape[0]), dtype=torch.float64, device=dev)
    onehot[torch.from_numpy(ao_atom).to(dev), torch.arange(d.shape[0], device=dev)] = 1.0
    rows = torch.einsum("cxab,ab->cxa", ip, d)                              # sum over b, per bra AO
    out -= 2.0 * torch.einsum("ra,cxa->rx", onehot, rows)
    tot = rows.sum(dim=2)                                                   # [n_ecp, 3]
    for n, c in enumerate(ecp_atoms):
        out[c] += 2.0 * tot[n]
    return out


_AUX = {"p1": (1, 1), "m1": (-1, 0), "p2": (2, 2), "z0": (0, 1), "m2": (-2, 0)}      # kind -> (l shift, power of alpha in the coefficients)


def _aux_shells(packed, shells, kinds, first_ao):
    """Auxiliary shell rows for the derivative integrals: for every shell s and kind k a row with l + dl and coefficients
    alpha_p^n c_p, AOs appended after ``first_ao``.  Returns (rows, {(s, kind): (row index counted from len(packed), first AO)}, next AO)."""
    rows, where, off = [], {}, first_ao
    for s in shells:
        l, npr = int(packed[s, 11]), int(packed[s, 10])
        for kind in kinds:
            dl, na = _AUX[kind]
            if l + dl < 0:
                continue
            r = packed[s].copy()
            r[11], r[3] = l + dl, off
            r[4:4 + 2 * npr:2] = packed[s, 4:4 + 2 * npr:2] * packed[s, 5:5 + 2 * npr:2] ** na
            where[(int(s), kind)] = (len(rows), off)
            rows.append(r)
            off += (l + dl + 1) * (l + dl + 2) // 2
    return rows, where, off


def _terms(l, comp, i, j=None):
    """d/dx_i (j None) or d^2/dx_i dx_j of the Cartesian Gaussian component ``comp`` (exponents) of an l shell as a list of
    (weight, kind, exponents of the auxiliary shell's component)."""
    n = list(comp)
    sh = lambda d: tuple(n[x] + d[x] for x in range(3))
    e = lambda x, v: tuple(v if y == x else 0 for y in range(3))
    if j is None:
        out = [(-2.0, "p1", sh(e(i, 1)))]
        if n[i] > 0:
            out.append((float(n[i]), "m1", sh(e(i, -1))))
        return out
    if i == j:
        out = [(4.0, "p2", sh(e(i, 2))), (-2.0 * (2 * n[i] + 1), "z0", tuple(n))]
        if n[i] > 1:
            out.append((float(n[i] * (n[i] - 1)), "m2", sh(e(i, -2))))
        return out
    pp = tuple(n[x] + (x == i) + (x == j) for x in range(3))
    out = [(4.0, "p2", pp)]
    if n[j] > 0:
        out.append((-2.0 * n[j], "z0", tuple(n[x] + (x == i) - (x == j) for x in range(3))))
    if n[i] > 0:
        out.append((-2.0 * n[i], "z0", tuple(n[x] - (x == i) + (x == j) for x in range(3))))
    if n[i] > 0 and n[j] > 0:
        out.append((float(n[i] * n[j]), "m2", tuple(n[x] - (x == i) - (x == j) for x in range(3))))
    return out


def get_ecp_ipip(mol_or_basis_layout, ip_type="ipipv", ecp_atoms=None, precision="fp64", nr=NR_DEFAULT):
    """``[n_ecp_atoms, 9, nao, nao]`` (component 3 i + j): "ipipv" = <d_i d_j a| U_C |b>, "ipvip" = <d_i a| U_C |d_j b>, per ECP atom C
    (reference ``get_ecp_ipip``, backend/ecp.py:1141-1340; libcint's ``ECPscalar_ipiprinv`` / ``ECPscalar_iprinvip``), from the value
    kernel on auxiliary shells (module docstring); g shells need its l <= 6 instantiation."""
    import torch
    if ip_type not in ("ipipv", "ipvip"):
        raise ValueError(f"Invalid ip_type: {ip_type}. Supported types: 'ipipv', 'ipvip'")
    if precision != "fp64":
        raise ValueError("Only double precision ('fp64') is supported for ECP kernels")
    if hasattr(mol_or_basis_layout, "packed"):
        layout, mol = mol_or_basis_layout, mol_or_basis_layout._mol
    else:
        from ..pyscf.basis import BasisLayout
        mol = mol_or_basis_layout
        layout = BasisLayout.from_mol(mol, alignment=1)
    dev = _lib.require_gpu()
    nao_mol = mol.nao
    have = getattr(mol, "_ecpbas", None) is not None and len(mol._ecpbas) > 0
    all_atoms = sorted({int(a) for a in np.asarray(mol._ecpbas)[:, 0]}) if have else []
    want = all_atoms if ecp_atoms is None else [a for a in ecp_atoms if a in all_atoms]
    if not want:
        return torch.zeros((0 if ecp_atoms is None else len(ecp_atoms), 9, nao_mol, nao_mol), dtype=torch.float64, device=dev)
    xyz, loc, terms = ecp_arrays(mol)
    packed = np.asarray(layout.packed)
    shells = [int(s) for s in np.nonzero(~np.asarray(layout.pad_id))[0]]
    assert max(int(packed[s, 11]) for s in shells) <= 4, "ECP kernels: shells up to l = 4"
    nao = int(layout.nao)
    bra_kinds = ("p2", "z0", "m2") if ip_type == "ipipv" else ("p1", "m1")
    rows_a, where_a, off = _aux_shells(packed, shells, bra_kinds, nao)
    rows_b, where_b, off = _aux_shells(packed, shells, ("p1", "m1") if ip_type == "ipvip" else (), off)
    ntot = off
    table = np.ascontiguousarray(np.stack([packed[s] for s in range(packed.shape[0])] + rows_a + rows_b))
    base_a, base_b = packed.shape[0], packed.shape[0] + len(rows_a)
    lmax_shell = int(table[:, 11].max())
    cidx = {l: {c: n for n, c in enumerate(_cart(l))} for l in range(0, 7)}

    def side(where, i=None, j=None):
        """per Cartesian AO row of the internal basis: list of (weight, source row in the combined AO space) for d_i (d_j)"""
        lists = [[] for _ in range(nao)]
        for s in shells:
            l, a0 = int(packed[s, 11]), int(packed[s, 3])
            for n, c in enumerate(_cart(l)):
                for wgt, kind, ex in _terms(l, c, i, j):
                    lists[a0 + n].append((wgt, where[(s, kind)][1] + cidx[l + _AUX[kind][0]][ex]))
        width = max(len(x) for x in lists)
        idx = np.full((width, nao), ntot, dtype=np.int64)                 # (ntot: a row / column of zeros)
        wts = np.zeros((width, nao))
        for r, x in enumerate(lists):
            for t, (wgt, src) in enumerate(x):
                idx[t, r], wts[t, r] = src, wgt
        return torch.from_numpy(idx).to(dev), torch.from_numpy(wts).to(dev)

    r, w = radial_grid(nr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tab_d, loc_d, terms_d, xyz_d, r_d, w_d, ylm_d = t(table), t(loc), t(terms), t(xyz), t(r), t(w), t(ylm_table())
    amin_of = np.zeros(packed.shape[0])
    for s in shells:
        amin_of[s] = packed[s, 5:5 + 2 * int(packed[s, 10]):2].min()
    bra_rows = [(base_a + where_a[key][0], key[0]) for key in where_a]
    ket_rows = [(base_b + where_b[key][0], key[0]) for key in where_b] if ip_type == "ipvip" else [(s, s) for s in shells]
    ecp_index = {a: k for k, a in enumerate(all_atoms)}
    out = torch.zeros((len(want), 9, nao_mol, nao_mol), dtype=torch.float64, device=dev)
    if ip_type == "ipipv":
        sides = {(i, j): side(where_a, i, j) for i in range(3) for j in range(3)}
    else:
        sa = {i: side(where_a, i) for i in range(3)}
        sb = {j: side(where_b, j) for j in range(3)}
    for n, atom in enumerate(want):
        k = ecp_index[atom]
        z = float(terms[loc[k]:loc[k + 1], 2].min())
        tasks = _screened_pairs(np.asarray([ra for ra, _ in bra_rows]), np.asarray([pa for _, pa in bra_rows]),
                                np.asarray([rb for rb, _ in ket_rows]), np.asarray([pb for _, pb in ket_rows]), packed, amin_of, xyz[k], z, k)
        mat = torch.zeros((ntot + 1, ntot + 1), dtype=torch.float64, device=dev)
        if len(tasks):
            tk = t(np.asarray(tasks, dtype=np.int32))
            _lib.check(_lib.lib().jqc_ecp_scalar(tab_d.data_ptr(), ntot + 1, tk.data_ptr(), int(tasks.shape[0]), xyz_d.data_ptr(), loc_d.data_ptr(),
                                                 terms_d.data_ptr(), r_d.data_ptr(), w_d.data_ptr(), int(nr), ylm_d.data_ptr(),
                                                 mat.data_ptr(), 0, lmax_shell, _lib.stream_ptr()))
        comps = []
        for i in range(3):
            for j in range(3):
                if ip_type == "ipipv":
                    ia, wa = sides[(i, j)]
                    comps.append(sum(wa[q][:, None] * mat[ia[q], :nao] for q in range(ia.shape[0])))
                else:
                    (ia, wa), (ib, wb) = sa[i], sb[j]
                    acc = 0
                    for q in range(ia.shape[0]):
                        for p in range(ib.shape[0]):
                            acc = acc + (wa[q][:, None] * wb[p][None, :]) * mat[ia[q]][:, ib[p]]
                    comps.append(acc)
        out[n] = layout.dm_to_mol(torch.stack(comps))
        torch.cuda.current_stream().synchronize()
    return out
