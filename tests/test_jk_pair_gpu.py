"""GPU parity of the pair-based J/K backend (joltqc_amd/pyscf/jk_pair.py) against the CPU oracle, modelled on the
reference's jqc/pyscf/tests/test_jk_pair.py:63-290 (double precision, several density matrices, J only / K only,
range separation, well-separated atoms, spherical basis; tolerance 1e-7 there, 1e-9 here), plus the cross-check the
scheme-table gates use elsewhere: pair == tile == oracle class by class."""
import numpy as np
import pytest

from conftest import H2O, benzene_atoms

pytestmark = pytest.mark.gpu


def _setup(atom, basis, cart=False, unit="angstrom"):
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import jk_pair
    from joltqc_amd.pyscf.basis import BasisLayout
    mol = mole.Mole(atom=atom, basis=basis, cart=cart, unit=unit)
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    return mol, lay, jk_pair.generate_jk_kernel(lay)


def _dm(nao, seed=9, n=None):
    np.random.seed(seed)
    if n is None:
        d = np.random.rand(nao, nao)
        return d @ d.T
    d = np.random.rand(n, nao, nao)
    return np.einsum("nij,nkj->nik", d, d)


def _np(x):
    return x.detach().cpu().numpy()


@pytest.mark.parametrize("cart", [True, False])
def test_jk_pair_double(cart):
    from oracle import dense
    mol, lay, get_jk = _setup(H2O, "def2-tzvpp", cart)
    dm = _dm(mol.nao)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    sc = max(np.abs(rj).max(), np.abs(rk).max())
    assert vj.shape == dm.shape and np.abs(_np(vj) - rj).max() < 1e-11 * sc and np.abs(_np(vk) - rk).max() < 1e-11 * sc
    assert get_jk.stats["pair_classes"] > 30            # s..f of this basis: most class pairs have a pair kernel
    assert int(get_jk.stats["pair_counter"].item()) > 0


def test_jk_pair_multiple_dm_j_only_k_only_nonsymmetric():
    from joltqc_amd.pyscf import jk_pair
    from oracle import dense
    mol, lay, get_jk = _setup(H2O, "def2-svp")
    dm = _dm(mol.nao, n=3)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    assert vj.shape == dm.shape and np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9
    vj1 = jk_pair.generate_get_j(lay)(mol, dm[0], hermi=1)
    vk1 = jk_pair.generate_get_k(lay)(mol, dm[0], hermi=1)
    assert np.abs(_np(vj1) - rj[0]).max() < 1e-9 and np.abs(_np(vk1) - rk[0]).max() < 1e-9
    np.random.seed(3)
    dn = np.random.rand(mol.nao, mol.nao)               # non-symmetric density, hermi = 0
    vj0, vk0 = get_jk(mol, dn, hermi=0)
    rj0, rk0 = dense.get_jk(lay, dn, hermi=0)
    assert np.abs(_np(vj0) - rj0).max() < 1e-9 and np.abs(_np(vk0) - rk0).max() < 1e-9


def test_jk_pair_omega_and_screening():
    from oracle import dense
    mol, lay, get_jk = _setup(H2O, "def2-svp")
    dm = _dm(mol.nao)
    for omega in (0.3, 0.5):
        vj, vk = get_jk(mol, dm, hermi=1, omega=omega)
        rj, rk = dense.get_jk(lay, dm, hermi=1, omega=omega)
        assert np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9
    mol, lay, get_jk = _setup("H 0 0 0; H 0 0 100", "def2-tzvpp", unit="B")      # reference test_jk_pair_screening
    dm = _dm(mol.nao)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    assert np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9
    assert int(get_jk.stats["pair_counter"].item()) < (lay.nbasis * (lay.nbasis + 1) // 2) ** 2


def test_pair_j_equals_tile_j_class_by_class_benzene_svp():
    """The gate of the second algorithm: J from the pair kernels == J from the tiled kernels == oracle, at full size on
    benzene / def2-SVP (every s..d class pair), and on an s..f system where the largest classes (their pair kernel would
    spill registers) fall back to the tiled J kernels."""
    from joltqc_amd.pyscf import jk as jkmod
    from oracle import dense
    mol, lay, get_jk = _setup(benzene_atoms(), "def2-svp")
    dm = _dm(mol.nao)
    vj, _ = get_jk(mol, dm, hermi=1, with_k=False)
    tj, _ = jkmod.generate_jk_kernel(lay)(mol, dm, hermi=1, with_k=False)
    rj, _ = dense.get_jk(lay, dm, hermi=1, with_k=False)
    sc = np.abs(rj).max()
    assert np.abs(_np(vj) - rj).max() < 1e-11 * sc and np.abs(_np(vj) - _np(tj)).max() < 1e-11 * sc
    assert get_jk.stats["tile_classes"] == 0
    shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
              [2, [0.8, 1.0]], [3, [0.9, 1.0]]]
    mol, lay, get_jk = _setup("C 0 0 0; C 0 0.3 2.4; H 1.5 0.2 0.9", {"C": shells, "H": shells}, unit="B")
    dm = _dm(mol.nao)
    vj, _ = get_jk(mol, dm, hermi=1, with_k=False)
    rj, _ = dense.get_jk(lay, dm, hermi=1, with_k=False)
    assert np.abs(_np(vj) - rj).max() < 1e-11 * np.abs(rj).max()
    assert get_jk.stats["tile_classes"] > 0 and get_jk.stats["pair_classes"] > 0
