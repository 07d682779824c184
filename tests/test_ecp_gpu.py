"""Device ECP integrals against the brute-force oracle, through the C ABI (jqc_ecp_scalar).

Molecules, basis and potentials are the reference's own (jqc/pyscf/tests/test_ecp_small.py:28-93: Na2, s/p/d + one g function with
general contractions; type 1 = local channel only, type 2 = S, P and G projectors), spherical and Cartesian like its
``test_ecp_type{1,2}_{sph,cart}``.  The reference's bar is |h_gpu - h_libcint| < 1e-6 in the Frobenius norm; libcint is absent
here, so the comparison is with oracle/ecp.py (PARITY UNPINNED), at 1e-8 of the largest element."""
import numpy as np
import pytest

from test_ecp_oracle import ECP_TYPE1, ECP_TYPE2, na2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cart", [False, True])
@pytest.mark.parametrize("kind", ["type1", "type2"])
def test_get_ecp_against_the_oracle(kind, cart):
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import ecp as oecp
    mol = na2(ECP_TYPE1 if kind == "type1" else ECP_TYPE2, cart=cart)
    lay = BasisLayout.from_mol(mol, alignment=1)
    ref = oecp.ecp_scalar_mol(lay, mol, nang=32, nrad=32)
    got = becp.get_ecp(mol).cpu().numpy()
    assert got.shape == (mol.nao, mol.nao)
    scale = np.abs(ref).max()
    assert np.abs(got - got.T).max() < 1e-12 * scale
    assert np.abs(got - ref).max() < 1e-8 * scale, np.abs(got - ref).max() / scale
    assert np.linalg.norm(got - ref) < 1e-6                       # the reference's own criterion (Frobenius norm)
    # the radial grid is converged: 128 and 192 points give the same matrix
    for nr in (128, 192):
        other = becp.get_ecp(mol, nr=nr).cpu().numpy()
        assert np.abs(other - got).max() < 1e-10 * scale, (nr, np.abs(other - got).max() / scale)


def test_both_channel_types_far_and_near_the_centre():
    """Local + semi-local channels on ONE atom of a heteronuclear pair with tight and diffuse shells, a 1/r and a 1/r^2 term:
    shells on the ECP centre (kappa = 0 branch of the Bessel functions), on the neighbour 1.1 and 3 Bohr away (series and
    upward-recurrence branches: kappa = 2 alpha r |A - C| passes 16)."""
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import ecp as oecp
    bas = {"Na": [[0, [35.0, 0.3], [4.0, 0.5], [0.6, 0.4]], [1, [2.2, 1.0]], [2, [0.9, 1.0]], [3, [1.1, 1.0]]],
           "H": [[0, [6.0, 0.4], [0.9, 0.7]], [1, [1.4, 1.0]], [2, [1.0, 1.0]], [4, [1.3, 1.0]]]}
    text = "Na nelec 10\nNa ul\n1 2.1 -1.5\n2 0.9 0.7\n0 3.0 0.4\nNa S\n2 5.0 40.0\n0 1.2 2.0\nNa P\n2 3.1 12.0\nNa D\n1 1.7 -3.0\nNa F\n2 1.1 1.5\n"
    for dist in (1.1, 3.0):
        mol = mole.Mole(atom=f"Na 0.1 -0.2 0.3; H {0.1 + dist * 0.6} {-0.2 + dist * 0.0} {0.3 + dist * 0.8}", basis=bas,
                        ecp={"Na": text}, unit="B")
        lay = BasisLayout.from_mol(mol, alignment=1)
        ref = oecp.ecp_scalar_mol(lay, mol, nang=64, nrad=48)
        got = becp.get_ecp(lay).cpu().numpy()
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() < 1e-8 * scale, (dist, np.abs(got - ref).max() / scale)
        # block by block (the potential is short-ranged: at 3 Bohr the H-H block is 1e-4 of the largest element; the oracle's own
        # angular quadrature is converged to 2e-9 of that block there, 1e-14 elsewhere)
        n1 = 1 + 3 + 5 + 7
        for name, sl in (("NaNa", (slice(0, n1), slice(0, n1))), ("NaH", (slice(0, n1), slice(n1, None))), ("HH", (slice(n1, None), slice(n1, None)))):
            blk = np.abs(ref[sl]).max()
            assert np.abs(got[sl] - ref[sl]).max() < 1e-7 * blk, (dist, name, np.abs(got[sl] - ref[sl]).max() / blk)


@pytest.mark.parametrize("cart", [False, True])
@pytest.mark.parametrize("kind", ["type1", "type2"])
def test_get_ecp_ip_against_the_oracle(kind, cart):
    """First derivatives (reference get_ecp_ip, backend/ecp.py:953-1138; its tests: test_ecp_small.py:133-147 per ECP atom against
    libcint's ECPscalar_iprinv): <grad a| U_C |b> for each ECP atom, assembled from l + 1 / l - 1 auxiliary bra shells through the
    value kernel, against the oracle's quadrature with the AO gradients; spherical AND Cartesian on the device (the reference
    computes the spherical case on the CPU)."""
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import ecp as oecp
    mol = na2(ECP_TYPE1 if kind == "type1" else ECP_TYPE2, cart=cart)
    lay = BasisLayout.from_mol(mol, alignment=1)
    got = becp.get_ecp_ip(mol).cpu().numpy()
    assert got.shape == (2, 3, mol.nao, mol.nao)
    for n, atom in enumerate((0, 1)):
        ref = oecp.ecp_ip_mol(lay, mol, atom, nang=32, nrad=32)
        scale = np.abs(ref).max()
        assert np.abs(got[n] - ref).max() < 1e-8 * scale, (atom, np.abs(got[n] - ref).max() / scale)
    one = becp.get_ecp_ip(mol, ecp_atoms=[1]).cpu().numpy()
    assert one.shape == (1, 3, mol.nao, mol.nao) and np.abs(one[0] - got[1]).max() < 1e-13 * np.abs(got[1]).max()
    with pytest.raises(ValueError):                          # (the reference's own argument check, ecp.py:969-971)
        becp.get_ecp_ip(mol, ip_type="ipipv")


@pytest.mark.parametrize("cart", [False, True])
@pytest.mark.parametrize("ip_type", ["ipipv", "ipvip"])
def test_get_ecp_ipip_against_the_oracle(ip_type, cart):
    """Second derivatives (reference get_ecp_ipip, backend/ecp.py:1141-1340; its tests: test_ecp_small.py:145-190, the sum over ECP atoms
    against libcint's ECPscalar_ipipnuc / ECPscalar_ipnucip): [n_ecp, 9, nao, nao] from the value kernel on l +- 2 / l +- 1 auxiliary
    shells (the g function of the reference's basis becomes an i shell: the kernel's l <= 6 instantiation), against the oracle's
    quadrature with analytic AO second derivatives."""
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import ecp as oecp
    mol = na2(ECP_TYPE2, cart=cart)
    lay = BasisLayout.from_mol(mol, alignment=1)
    got = becp.get_ecp_ipip(mol, ip_type=ip_type).cpu().numpy()
    assert got.shape == (2, 9, mol.nao, mol.nao)
    # (the oracle's AO Hessians make "ipipv" the slow leg: one ECP atom there, both for "ipvip")
    for n, atom in enumerate((0, 1) if ip_type == "ipvip" else (1,)):
        n = atom
        ref = oecp.ecp_ipip_mol(lay, mol, atom, ip_type, nang=32, nrad=32)
        scale = np.abs(ref).max()
        assert np.abs(got[n] - ref).max() < 1e-8 * scale, (atom, np.abs(got[n] - ref).max() / scale)
    g9 = got.reshape(2, 3, 3, mol.nao, mol.nao)
    if ip_type == "ipipv":
        assert np.abs(g9 - g9.transpose(0, 2, 1, 3, 4)).max() < 1e-12 * np.abs(g9).max()
    else:
        assert np.abs(g9 - g9.transpose(0, 2, 1, 4, 3)).max() < 1e-12 * np.abs(g9).max()
    one = becp.get_ecp_ipip(mol, ip_type=ip_type, ecp_atoms=[1]).cpu().numpy()
    assert one.shape == (1, 9, mol.nao, mol.nao) and np.abs(one[0] - got[1]).max() < 1e-13 * np.abs(got[1]).max()
    with pytest.raises(ValueError):                          # (the reference's own argument check, ecp.py:1158-1161)
        becp.get_ecp_ipip(mol, ip_type="ip")


def test_ecp_force_term_against_finite_differences():
    """d/dR tr(D h_ECP) at fixed D from the first-derivative blocks (translational invariance for the ECP centre itself)
    against central differences of the device's own value integrals, every atom and direction of a bent three-atom molecule
    with one ECP atom."""
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.gto import mole
    bas = {"Na": [[0, [4.0, 0.5], [0.6, 0.6]], [1, [1.2, 1.0]], [2, [0.8, 1.0]]], "H": [[0, [1.1, 1.0]], [1, [0.9, 1.0]]]}
    text = "Na nelec 10\nNa ul\n2 0.9 0.7\n1 1.8 -1.1\nNa S\n2 3.0 25.0\nNa P\n2 2.1 9.0\n"
    xyz = np.array([[0.0, 0.1, -0.2], [2.3, 0.4, 0.3], [-0.9, 2.0, 0.8]])
    build = lambda r: mole.Mole(atom=[("Na", r[0]), ("H", r[1]), ("H", r[2])], basis=bas, ecp={"Na": text}, unit="B")
    mol = build(xyz)
    rng = np.random.default_rng(3)
    D = rng.random((mol.nao, mol.nao)) - 0.5
    D = D + D.T
    g = becp.ecp_energy_per_atom(mol, D).cpu().numpy()
    energy = lambda r: float((becp.get_ecp(build(r)).cpu().numpy() * D).sum())
    h = 2e-4
    fd = np.zeros((3, 3))
    for a in range(3):
        for x in range(3):
            rp, rm = xyz.copy(), xyz.copy()
            rp[a, x] += h
            rm[a, x] -= h
            fd[a, x] = (energy(rp) - energy(rm)) / (2 * h)
    assert np.abs(g.sum(0)).max() < 1e-10 * np.abs(g).max()               # translational invariance
    assert np.abs(g - fd).max() < 1e-6 * np.abs(fd).max(), (g, fd)


def test_task_screening_changes_nothing():
    """A chain of four ECP atoms 7 Bohr apart: the distance screening of the (shell pair, ECP atom) tasks (the reference keeps
    every triple) drops most of them and leaves the matrix unchanged to 1e-14 of its largest element."""
    from joltqc_amd.backend import ecp as becp
    from joltqc_amd.gto import mole
    from test_ecp_oracle import BAS
    mol = mole.Mole(atom="; ".join(f"Na {7.0 * n} {0.3 * n} 0" for n in range(4)), basis={"Na": BAS}, ecp={"Na": ECP_TYPE2}, unit="B")
    full = becp.get_ecp(mol, screen=False).cpu().numpy()
    n_full = becp.get_ecp.last_ntasks
    cut = becp.get_ecp(mol).cpu().numpy()
    assert becp.get_ecp.last_ntasks < 0.7 * n_full, (becp.get_ecp.last_ntasks, n_full)
    assert np.abs(cut - full).max() < 1e-14 * np.abs(full).max()


def test_rhf_with_ecp_through_apply_with_device_integrals():
    """The row in use: RHF on Na2 with the reference's type-2 potential (two valence electrons) through ``apply()`` with
    ``int1e=True`` -- overlap, kinetic energy, nuclear attraction of the lowered charges AND the ECP matrix from the device, J / K
    from the tiled kernels up to (gg|gg) -- against the same SCF with every integral from the CPU oracles."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    from oracle import ecp as oecp
    from standin_scf import RHF
    mol = na2(ECP_TYPE2)
    assert mol.has_ecp() and mol.nelectron == 2
    lay = BasisLayout.from_mol(mol, alignment=1)
    S, T, V = dense.int1e_mol(lay, mol)
    h_ref = T + V + oecp.ecp_scalar_mol(lay, mol, nang=32, nrad=32)
    ref = RHF(mol, h_ref, S)
    q = dense.canonical_quartets(lay)
    ref.get_jk = lambda mol_=None, dm=None, hermi=1, **kw: dense.get_jk(lay, dm, hermi, quartets=q)
    e_ref = ref.kernel()
    cfg = jp.get_default_config()
    cfg["int1e"] = True
    mf = jp.apply(RHF(mol, None, None), cfg)
    mf._hcore, mf._ovlp = mf.get_hcore(), mf.get_ovlp()
    assert np.abs(mf._hcore - h_ref).max() < 1e-8 * np.abs(h_ref).max()
    e = mf.kernel()
    assert ref.converged and mf.converged and abs(e - e_ref) < 1e-8, (e, e_ref)


def test_patch_interface_mirrors_the_reference():
    """jqc/pyscf/ecp.py:27-118: apply_ecp -> dict of closures, patch_ecp_integrals installs mol.get_ecp, restore removes it; a
    molecule without ECP is left alone."""
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import ecp as pecp
    mol = na2(ECP_TYPE1)
    patches = pecp.apply_ecp(mol)
    assert set(patches) == {"get_ecp", "ecp_kernel", "precision", "original_methods"} and patches["precision"] == "fp64"
    with pytest.raises(ValueError):
        pecp.apply_ecp(mol, precision="fp32")
    pecp.patch_ecp_integrals(mol)
    h = mol.get_ecp().cpu().numpy()
    assert h.shape == (mol.nao, mol.nao) and np.abs(h).max() > 0.1 and mol._jqc_ecp_info["precision"] == "fp64"
    pecp.restore_ecp_methods(mol)
    assert not hasattr(mol, "_jqc_ecp_info") and "get_ecp" not in mol.__dict__
    plain = mole.Mole(atom="H 0 0 0; H 0 0 1.4", basis="def2-svp", unit="B")
    assert pecp.apply_ecp(plain) == {}
