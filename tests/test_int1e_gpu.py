"""jqc_int1e (overlap, kinetic, nuclear attraction on the device) against the independent
McMurchie-Davidson engine of the oracle, s..g, contracted and split shells, Cartesian and spherical molecules; and the
reference's hard-coded H2O / def2-TZVPP RHF energy (jqc/pyscf/tests/test_scf.py:70) with EVERY integral from the device."""
import numpy as np
import pytest

from conftest import H2O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cart", [True, False])
def test_int1e_against_the_md_engine(cart):
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import int1e
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
              [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
    mol = mole.Mole(atom="C 0 0 0; O 0 0.3 2.4; H 1.5 0.2 0.9", basis={"C": shells, "O": "def2-tzvpp", "H": shells}, unit="B", cart=cart)
    lay = BasisLayout.from_mol(mol)
    S, T, V = (x.cpu().numpy() for x in int1e.int1e(lay, mol))
    rS, rT, rV = dense.int1e_mol(lay, mol)
    assert S.shape == (mol.nao, mol.nao)
    for got, ref in ((S, rS), (T, rT), (V, rV)):
        assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
        assert np.abs(got - got.T).max() < 1e-12 * max(1.0, np.abs(ref).max())


def test_rhf_energy_with_every_integral_from_the_device(kats):
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from standin_scf import RHF
    k = kats["h2o_def2tzvpp"]
    mol = mole.Mole(atom=k["atom"], basis="def2-tzvpp")
    mf = RHF(mol, int1e=lambda m: (None, None))                # no CPU integrals at all
    mf = jp.apply(mf, {**jp.get_default_config(), "int1e": True})
    mf._hcore, mf._ovlp = mf.get_hcore(), mf.get_ovlp()
    assert isinstance(mf._hcore, np.ndarray)
    e = mf.kernel()
    assert mf.converged and abs(e - k["e_rhf_sph"]) < 1e-8, e - k["e_rhf_sph"]
