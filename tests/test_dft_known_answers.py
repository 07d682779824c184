"""The reference's own DFT energies pin the grid-path oracle: H2O / def2-TZVPP with "LDA,vwn5" and "PBE"
(/root/reference/jqc/pyscf/tests/test_dft.py:75-86, e_ref -75.9046410402 / -76.3800182418, tolerance 1e-5 there).

Everything on the path is this repo's CPU restatement: oracle/dft.py (AO values, rho, V_xc), oracle/xc.py (closed-form
functionals), oracle/rks.py (nr_rks / get_veff), the Rys J oracle, the MD one-electron integrals and the Becke grid
of joltqc_amd/gto/grids.py (grid generation is third party in the reference: PySCF level 5 there, a product grid here,
refined until the energy is stable).  The same energies through ``apply()`` on the GPU: tests/test_dft_gpu.py.
"""
import numpy as np
import pytest

from joltqc_amd.gto import grids as G
from joltqc_amd.gto import mole
from joltqc_amd.pyscf.basis import BasisLayout
from oracle import dense, rks
from standin_scf import RKS, Grids

E_REF = {"lda,vwn5": -75.9046410402, "pbe": -76.3800182418}


def oracle_rks_energy(kats, xc_code, nrad, ntheta):
    mol = mole.Mole(atom=kats["h2o_def2tzvpp"]["atom"], basis="def2-tzvpp")
    lay = BasisLayout.from_mol(mol)
    S, T, V = dense.int1e_mol(lay, mol)
    q = dense.canonical_quartets(lay)
    g = G.Grids(mol, nrad=nrad, ntheta=ntheta).build()
    mf = RKS(mol, T + V, S, Grids(g.coords, g.weights), xc=xc_code)
    mf.get_veff = rks.make_get_veff(lay, g.coords, g.weights, xc_code,
                                    lambda dm: dense.get_jk(lay, dm, 1, with_k=False, quartets=q)[0])
    e = mf.kernel()
    assert mf.converged
    return e, mf.get_veff.stats["nelec"]


@pytest.mark.parametrize("xc_code", ["lda,vwn5", "pbe"])
def test_h2o_def2tzvpp_rks_energy_matches_the_reference(kats, xc_code):
    # grid refined until the energy is stable: (60, 16) -> (90, 24) moves it by 9e-8 / 4e-7 Eh, (90, 24) sits 1e-9 / 6e-8 Eh
    # from the reference's number (which carries PySCF's own level-5 grid error); the bar is ten times tighter than the
    # reference's own 1e-5
    e, nelec = oracle_rks_energy(kats, xc_code, 90, 24)
    assert abs(nelec - 10.0) < 1e-7, nelec
    assert abs(e - E_REF[xc_code]) < 1e-6, e - E_REF[xc_code]
