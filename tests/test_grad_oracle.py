"""CPU: the analytic gradient oracle of the two-electron energy (oracle/grad.py, independent McMurchie-Davidson engine) against
finite differences of the PINNED J/K oracle (oracle/dense.py -> jk_oracle.c).  The reference has no gradient code to compare
with; this is what pins the checker the GPU tests use."""
import numpy as np
import pytest

from joltqc_amd.gto import mole
from joltqc_amd.pyscf.basis import BasisLayout
from oracle import grad as G

BASIS = {"O": [[0, [11.0, 0.3], [2.1, 0.5], [0.6, 0.4]], [1, [1.9, 0.6], [0.45, 0.5]], [2, [0.9, 1.0]]],
         "H": [[0, [3.4, 0.2], [0.6, 0.8]], [1, [0.8, 1.0]]]}
COORDS = np.array([[0.0, 0.05, -0.1], [0.3, 1.1, 1.45]])


def _layout(c, cart=False):
    mol = mole.Mole(atom=[("O", tuple(c[0])), ("H", tuple(c[1]))], basis=BASIS, unit="B", cart=cart)
    return BasisLayout.from_mol(mol, alignment=1)


@pytest.mark.parametrize("case", ["rhf", "uhf_lr"])
def test_analytic_gradient_oracle_matches_finite_differences(case):
    lay = _layout(COORDS)
    rng = np.random.default_rng(5)
    n = lay.nao_mol
    if case == "rhf":
        d = rng.random((n, n)) - 0.3
        dm, jf, kf, om = d + d.T, 1.0, 1.0, None
    else:
        a, b = rng.random((n, n)) - 0.4, rng.random((n, n)) - 0.6
        dm, jf, kf, om = np.stack([a + a.T, b + b.T]), 0.7, 0.35, 0.4
    ana = G.jk_energy_per_atom(lay, dm, jf, kf, om)
    fd = G.jk_energy_per_atom_fd(_layout, COORDS, dm, jf, kf, om)
    assert np.abs(ana.sum(0)).max() < 1e-9 * np.abs(ana).max()              # translational invariance
    assert np.abs(ana - fd).max() < 2e-8 * np.abs(fd).max(), (ana, fd)
