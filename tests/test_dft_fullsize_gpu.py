"""Grid path at BASELINE size (config 3: Taxol-size molecule, def2-SVP) on the GPU: size-independent properties on a real
Becke quadrature grid -- int rho = N_e, V_xc symmetric, tr(V M) = int (wv . rho_M) (the rho and vxc kernels are each other's
adjoint), sampled 256-point blocks of rho against the NumPy oracle -- with all-FP64 cutoffs and with the default mixed
FP32/FP64 windows.  Every integral (S, hcore for the density) comes from the device kernels."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_112_atoms_svp_grid_path_properties():
    import torch
    from joltqc_amd.gto import grids as G, mole
    from joltqc_amd.pyscf import int1e, rks
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dft
    mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules/0112-elongated-nitrogenous.xyz")),
                    basis="def2-svp")
    lay = BasisLayout.from_mol(mol, alignment=1)
    S, T, V = (x.cpu().numpy() for x in int1e.int1e(lay, mol))
    w, U = np.linalg.eigh(S)
    X = U[:, w > 1e-9] / np.sqrt(w[w > 1e-9])
    _, c = np.linalg.eigh(X.T @ (T + V) @ X)
    nocc = mol.nelectron // 2
    C = X @ c[:, :nocc]
    D = 2.0 * C @ C.T                                   # core-Hamiltonian density: tr(D S) = N_e exactly
    assert abs(np.trace(D @ S) - mol.nelectron) < 1e-8

    g = G.Grids(mol, 50, 10).build()
    order = rks.arg_group_grids(g.coords)
    g.coords, g.weights = g.coords[order], g.weights[order]
    npad = (-len(g.weights)) % 256
    g.coords = np.vstack([g.coords, np.repeat(g.coords[-1:], npad, axis=0)])
    g.weights = np.concatenate([g.weights, np.zeros(npad)])
    n = len(g.weights)
    assert n > 5e5

    rng = np.random.default_rng(3)
    wv = rng.random((4, n)) * g.weights
    M = rng.random((mol.nao, mol.nao)) - 0.5
    M = M + M.T
    results = {}
    for label, c64 in (("fp64", 1e-13), ("mixed", 1e-6)):
        _, rho_k, vxc_k = rks.generate_rks_kernel(lay, cutoff_fp64=c64, cutoff_fp32=1e-13)
        rho = rho_k(mol, g, "GGA", D).cpu().numpy()
        nelec = float((rho[0] * g.weights).sum())
        assert abs(nelec - mol.nelectron) < 2e-4 * mol.nelectron, (label, nelec)
        Vx = vxc_k(mol, g, "GGA", wv).cpu().numpy()
        assert np.abs(Vx - Vx.T).max() < 1e-12 * np.abs(Vx).max()
        rho_M = rho_k(mol, g, "GGA", M).cpu().numpy()
        lhs, rhs = float((Vx * M).sum()), float((wv * rho_M).sum())
        assert abs(lhs - rhs) < (1e-9 if label == "fp64" else 1e-6) * abs(rhs), (label, lhs, rhs)
        results[label] = (rho, nelec)
    # sampled blocks against the oracle (all AOs, no screening)
    rho = results["fp64"][0]
    for blk in (0, n // 512, n // 256 - 2):
        sl = slice(blk * 256, blk * 256 + 256)
        ref = dft.eval_rho(lay, g.coords[sl], D, "GGA")
        assert np.abs(rho[:, sl] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()), blk
    # mixed precision stays within the reference's 1e-7 of the all-FP64 density where it matters: the integral
    assert abs(results["mixed"][1] - results["fp64"][1]) < 1e-6 * mol.nelectron
