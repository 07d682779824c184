"""CPU checks of the ECP row: the input layer, the host-side tables of the device kernel and the
brute-force oracle (oracle/ecp.py) against closed forms.  PARITY UNPINNED: the reference's own tests compare with libcint's
``ECPscalar`` (jqc/pyscf/tests/test_ecp_small.py:118-131), which is third party and absent here, and store no numbers.
The molecules are the reference's: two Na atoms with its inline basis (s with three general contractions, p with two, d, and a
g function) and its type-1 (local channel) / type-2 (S, P and G projectors) potentials (test_ecp_small.py:28-93)."""
import math

import numpy as np
import pytest

BAS = [[4, [1.8, 1.0]],
       [0, [2.8, 0.021087, -0.00454, 0.0], [1.319, 0.346129, -0.170352, 0.0], [0.9059, 0.039378, 0.140382, 1.0]],
       [1, [2.133, 0.086866, 0.0], [1.2, 0.0, 0.5], [0.3827, 0.501008, 1.0]],
       [2, [0.3827, 1.0]]]
ECP_TYPE1 = "Na nelec 10\nNa ul\n2       1.0                   0.5\n"
ECP_TYPE2 = """Na nelec 10
Na S
2      13.652203             732.2692
2       6.826101              26.484721
Na P
2      10.279868             299.489474
2       5.139934              26.466234
Na G
2       7.349859             124.457595
2       3.674929              14.035995
"""
ATOM = "Na 0.5 0.5 0.; Na 0. 1. 1."


def na2(ecp_text, cart=False):
    from joltqc_amd.gto import mole
    return mole.Mole(atom=ATOM, basis={"Na": BAS}, ecp={"Na": ecp_text}, cart=cart)


def test_parse_ecp_builds_pyscf_rows():
    from joltqc_amd.gto import ecp as gecp
    mol = na2(ECP_TYPE2)
    assert mol._ecpbas.shape == (6, 8)                       # two atoms x three channels, one radial power each
    assert sorted(set(mol._ecpbas[:, gecp.ANG_OF].tolist())) == [0, 1, 4]
    assert (mol._ecpbas[:, gecp.NPRIM_OF] == 2).all() and (mol._ecpbas[:, gecp.RADI_POWER] == 2).all()
    assert mol.atom_charges().tolist() == [1, 1] and mol.nelectron == 2          # 10 core electrons per atom removed
    ch = gecp.channels(mol)
    l, power, zeta, coef = ch[0][0]
    assert (l, power) == (0, 2) and np.allclose(zeta, [13.652203, 6.826101]) and np.allclose(coef, [732.2692, 26.484721])
    m1 = na2(ECP_TYPE1)
    assert m1._ecpbas[:, gecp.ANG_OF].tolist() == [-1, -1]


def test_spin_orbit_rows_of_ecpbas_are_ignored():
    """A PySCF molecule whose ECP carries spin-orbit terms (crenbl / crenbs) has extra ``_ecpbas`` rows with SO_TYPE_OF != 0;
    the scalar potential must not see them (reference: /root/reference/jqc/backend/ecp.py:1313-1315 keeps SO_TYPE_OF == 0 only)."""
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.gto import ecp as gecp
    mol = na2(ECP_TYPE2)
    ref_ch = gecp.channels(mol)
    ref_xyz, ref_loc, ref_terms = becp.ecp_arrays(mol)
    so = mol._ecpbas[1].copy()                                # a P-channel row re-labelled as a spin-orbit term
    so[gecp.SO_TYPE_OF] = 1
    mol._ecpbas = np.vstack([mol._ecpbas[:2], so[None], mol._ecpbas[2:]]).astype(np.int32)
    assert mol._ecpbas.shape == (7, 8)
    ch = gecp.channels(mol)
    assert {k: len(v) for k, v in ch.items()} == {k: len(v) for k, v in ref_ch.items()}
    xyz, loc, terms = becp.ecp_arrays(mol)
    assert np.array_equal(loc, ref_loc) and np.array_equal(terms, ref_terms) and np.array_equal(xyz, ref_xyz)


def test_real_spherical_harmonic_table_is_orthonormal():
    """ylm_table (device input): int Y_lm Y_l'm' dOmega = delta, from the exact monomial integrals over the sphere."""
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.gto.c2s import cart_powers
    tab = becp.ylm_table(4)

    def dfact(n):
        return 1.0 if n <= 0 else float(np.prod(np.arange(n, 0, -2)))

    def mono(a, b, c):
        if a % 2 or b % 2 or c % 2:
            return 0.0
        return 4 * math.pi * dfact(a - 1) * dfact(b - 1) * dfact(c - 1) / dfact(a + b + c + 1)
    for l1 in range(5):
        for l2 in range(l1, 5):
            p1, p2 = cart_powers(l1), cart_powers(l2)
            S = np.array([[mono(a[0] + b[0], a[1] + b[1], a[2] + b[2]) for b in p2] for a in p1])
            G = tab[l1 * l1:(l1 + 1) ** 2, :len(p1)] @ S @ tab[l2 * l2:(l2 + 1) ** 2, :len(p2)].T
            assert np.abs(G - (np.eye(2 * l1 + 1) if l1 == l2 else 0.0)).max() < 1e-13, (l1, l2)


def test_radial_grid_integrates_gaussians():
    from joltqc_amd.backend import ecp as becp
    r, w = becp.radial_grid()
    for a, n in ((0.1, 2), (1.0, 2), (30.0, 4), (2000.0, 2), (0.3, 8), (5e4, 2), (13.6, 0), (13.6, 1), (800.0, 0), (5e4, 0)):
        exact = math.gamma((n + 1) / 2) / (2 * a ** ((n + 1) / 2))
        assert abs((w * r ** n * np.exp(-a * r * r)).sum() / exact - 1) < 1e-13, (a, n)


def test_oracle_against_closed_forms_on_the_ecp_centre():
    """One atom, everything centred on the ECP atom: <s|U_0 P_0|s> = <s|U_0|s> = sum c c' int r^2 R R' U dr (P_0 s = s), and a
    projector of another l annihilates an s function; the local channel on an s pair is the same radial integral."""
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import ecp as oecp
    bas = {"Na": [[0, [1.3, 1.0]], [0, [0.4, 1.0]], [1, [0.9, 1.0]]]}
    for chan in ("S", "ul", "P"):
        mol = mole.Mole(atom="Na 0 0 0", basis=bas, ecp={"Na": f"Na nelec 10\nNa {chan}\n2 1.1 3.0\n1 0.7 -0.8\n"})
        lay = BasisLayout.from_mol(mol, alignment=1)
        V = oecp.ecp_scalar_mol(lay, mol, nang=24, nrad=32)
        gi = lambda n, a: math.gamma((n + 1) / 2) / (2 * a ** ((n + 1) / 2))
        norm = lambda a: 1 / math.sqrt(gi(2, 2 * a))               # radial normalisation of an s primitive
        ex = lambda a, b: norm(a) * norm(b) * (3.0 * gi(2, a + b + 1.1) - 0.8 * gi(1, a + b + 0.7))
        want = np.array([[ex(1.3, 1.3), ex(1.3, 0.4)], [ex(0.4, 1.3), ex(0.4, 0.4)]])
        got = V[:2, :2]
        if chan == "P":
            assert np.abs(got).max() < 1e-13                      # a p projector sees no s function
            px = 3.0 * gi(4, 1.8 + 1.1) - 0.8 * gi(3, 1.8 + 0.7)
            assert abs(V[2, 2] - px / gi(4, 1.8)) < 1e-12
        else:
            assert np.abs(got - want).max() < 1e-12, (chan, got, want)
            assert np.abs(V[:2, 2:]).max() < 1e-13


@pytest.mark.parametrize("text", [ECP_TYPE1, ECP_TYPE2])
def test_oracle_is_converged_and_symmetric_on_the_reference_molecules(text):
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import ecp as oecp
    mol = na2(text)
    lay = BasisLayout.from_mol(mol, alignment=1)
    a = oecp.ecp_scalar_mol(lay, mol, nang=32, nrad=32)
    b = oecp.ecp_scalar_mol(lay, mol, nang=40, nrad=40)
    assert a.shape == (mol.nao, mol.nao) and np.abs(a - a.T).max() < 1e-12 * np.abs(a).max()
    assert np.abs(a - b).max() < 1e-12 * np.abs(b).max()


def test_oracle_first_derivative_against_finite_differences():
    """oracle.ecp.ecp_ip = <d/dr a| U_C |b>: moving the SHELLS of the other atom (the ECP centre stays) changes <a|U_C|b> by
    -<grad a|U_C|b> . dA for a on the moved atom, b on the fixed one."""
    import copy
    from joltqc_amd.gto import ecp as gecp
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import ecp as oecp
    mol = na2(ECP_TYPE2, cart=True)
    lay = BasisLayout.from_mol(mol, alignment=1)
    ch = {0: gecp.channels(mol)[0]}                               # the potential of atom 0 only
    xyz = mol.atom_coords()
    ip = oecp.ecp_ip(lay, ch, xyz, 0, nang=32, nrad=32)
    on1 = np.repeat(np.asarray(lay.atom_of) == 1, np.diff(lay.ao_loc))
    h = 1e-4
    for x in range(3):
        vs = []
        for sgn in (1, -1):
            l2 = copy.copy(lay)
            l2.packed = lay.packed.copy()
            l2.packed[np.asarray(lay.atom_of) == 1, x] += sgn * h
            vs.append(oecp.ecp_scalar(l2, ch, xyz, nang=32, nrad=32))
        fd = (vs[0] - vs[1]) / (2 * h)
        blk = np.ix_(on1, ~on1)
        assert np.abs(fd[blk] + ip[x][blk]).max() < 1e-7 * max(np.abs(ip[x][blk]).max(), 1e-3), x


def test_oracle_second_derivatives_against_finite_differences():
    """oracle.ecp.ao_second_derivatives against central differences of the AO gradients, and oracle.ecp.ecp_ipip against central
    differences of ecp_ip when the shells of the other atom move: d/dA_j <d_i a|U_C|b> = -<d_i d_j a|U_C|b> for a on the moved atom
    and b on the fixed one, = -<d_i a|U_C|d_j b> for a fixed and b moved."""
    import copy
    from joltqc_amd.gto import ecp as gecp
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dft as odft
    from oracle import ecp as oecp
    mol = na2(ECP_TYPE2, cart=True)
    lay = BasisLayout.from_mol(mol, alignment=1)
    pts = np.random.default_rng(3).normal(size=(40, 3)) * 1.5
    hess = oecp.ao_second_derivatives(lay.packed, lay.ao_loc, pts)
    h = 1e-5
    for j in range(3):
        d = np.zeros(3)
        d[j] = h
        fd = (odft.eval_ao_cart(lay.packed, lay.ao_loc, pts + d, deriv=1)[1:] - odft.eval_ao_cart(lay.packed, lay.ao_loc, pts - d, deriv=1)[1:]) / (2 * h)
        assert np.abs(fd - hess[:, j]).max() < 1e-8 * np.abs(hess).max(), j
    assert np.abs(hess - hess.transpose(1, 0, 2, 3)).max() == 0

    # (the quadrature grid sits on the fixed ECP centre, so the identity holds at any grid size: a coarse one keeps this fast)
    ch = {0: gecp.channels(mol)[0]}
    xyz = mol.atom_coords()
    ipipv = oecp.ecp_ipip(lay, ch, xyz, 0, "ipipv", nang=16, nrad=16)
    ipvip = oecp.ecp_ipip(lay, ch, xyz, 0, "ipvip", nang=16, nrad=16)
    assert np.abs(ipvip - ipvip.transpose(1, 0, 3, 2)).max() < 1e-12 * np.abs(ipvip).max()
    on1 = np.repeat(np.asarray(lay.atom_of) == 1, np.diff(lay.ao_loc))
    h = 1e-4
    for j in range(3):
        vs = []
        for sgn in (1, -1):
            l2 = copy.copy(lay)
            l2.packed = lay.packed.copy()
            l2.packed[np.asarray(lay.atom_of) == 1, j] += sgn * h
            vs.append(oecp.ecp_ip(l2, ch, xyz, 0, nang=16, nrad=16))
        fd = (vs[0] - vs[1]) / (2 * h)                            # [3 (i), nao, nao]
        moved_bra, moved_ket = np.ix_(on1, ~on1), np.ix_(~on1, on1)
        for i in range(3):
            assert np.abs(fd[i][moved_bra] + ipipv[i, j][moved_bra]).max() < 1e-6 * max(np.abs(ipipv[i, j][moved_bra]).max(), 1e-3), (i, j)
            assert np.abs(fd[i][moved_ket] + ipvip[i, j][moved_ket]).max() < 1e-6 * max(np.abs(ipvip[i, j][moved_ket]).max(), 1e-3), (i, j)
