"""Pins the DFT oracle's AO evaluation independently (analytic overlap + finite differences)."""
import numpy as np

from conftest import H2_BOHR
from joltqc_amd.gto import mole
from joltqc_amd.pyscf.basis import BasisLayout
from oracle import dense, dft

BASIS = {"H": [[0, [34.0613410, 0.60251978e-2], [5.1235746, 0.45021094e-1], [1.1646626, 0.20189726]],
               [0, [0.32723041, 1.0]], [1, [1.407, 1.0]], [1, [0.388, 1.0]], [2, [1.057, 1.0]], [3, [1.057, 1.0]]]}
# (the inline basis of the reference's tests/test_rks.py:37-56, minus its most diffuse s function)


def test_ao_values_reproduce_analytic_overlap():
    # valence part only: a uniform grid integrates Gaussians spectrally once h^2 * alpha << pi^2
    basis = {"H": [[0, [0.9, 1.0]], [1, [1.1, 1.0]], [2, [1.057, 1.0]], [3, [1.2, 1.0]], [4, [1.3, 1.0]]]}
    mol = mole.Mole(atom="H 0 0 0; H 0.3 -0.2 1.1", basis=basis, unit="B")
    lay = BasisLayout.from_mol(mol)
    S, _, _ = dense.int1e_mol(lay, mol)
    h = 0.22
    ax = np.arange(-7.0, 8.0, h)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    ao = dft.eval_ao_mol(lay, g)[0]
    Sq = (ao * h ** 3) @ ao.T
    assert np.abs(Sq - S).max() < 1e-9


def test_ao_gradients_match_finite_differences():
    mol = mole.Mole(atom=H2_BOHR, basis=BASIS, unit="B")
    lay = BasisLayout.from_mol(mol)
    rng = np.random.default_rng(0)
    g = rng.uniform(-2, 2, (50, 3)) + np.array([0.0, 4.0, -0.4696])
    ao = dft.eval_ao_mol(lay, g, deriv=1)
    d = 1e-5
    for x in range(3):
        e = np.zeros(3); e[x] = d
        fd = (dft.eval_ao_mol(lay, g + e)[0] - dft.eval_ao_mol(lay, g - e)[0]) / (2 * d)
        assert np.abs(fd - ao[1 + x]).max() < 1e-8


def test_rho_integrates_to_electron_count():
    # D = C C^T of one normalised MO: rho must integrate to <phi|phi> = 1
    basis = {"H": [[0, [0.9, 1.0]], [1, [1.1, 1.0]]]}
    mol = mole.Mole(atom="H 0 0 0; H 0 0 1.4", basis=basis, unit="B")
    lay = BasisLayout.from_mol(mol)
    S, _, _ = dense.int1e_mol(lay, mol)
    c = np.linalg.eigh(S)[1][:, -1]
    c /= np.sqrt(c @ S @ c)
    dm = np.outer(c, c)
    h = 0.25
    ax = np.arange(-8.0, 9.4, h)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    rho = dft.eval_rho(lay, g, dm, "MGGA")
    assert abs(rho[0].sum() * h ** 3 - 1.0) < 1e-9
    # tau integrates to the kinetic energy <phi|-1/2 lap|phi>
    _, T, _ = dense.int1e_mol(lay, mol)
    assert abs(rho[4].sum() * h ** 3 - c @ T @ c) < 1e-8
    # V_xc with w = weights reproduces the overlap (LDA form) and is symmetric in every form
    w = np.full(g.shape[0], h ** 3)
    assert np.abs(dft.eval_vxc(lay, g, w, "LDA") - S).max() < 1e-9
