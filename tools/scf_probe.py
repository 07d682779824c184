"""Convergence probe of the stand-in Kohn-Sham driver of the tests on the 112-atom molecule (B3LYP / def2-SVP through apply()):
per-iteration energy and DIIS error from the atomic-density guess (and, with --all, from the core-Hamiltonian guess with plain DIIS,
damping, a level shift: none of those converges in 50 cycles).
usage: python tools/scf_probe.py [max_cycle] [--all]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import joltqc_amd.pyscf as jp
from joltqc_amd.gto import mole
from joltqc_amd.gto.grids import Grids
from joltqc_amd.pyscf import int1e
from joltqc_amd.pyscf.basis import BasisLayout
from standin_scf import RKS, ClosedFormNumInt, _strict, atomic_density_guess

mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules/0112-elongated-nitrogenous.xyz")), basis="def2-svp")
S, T, V = (x.cpu().numpy() for x in int1e.int1e(BasisLayout.from_mol(mol, alignment=1), mol))
h = T + V
s, U = np.linalg.eigh(S)
X = U[:, s > 1e-10] / np.sqrt(s[s > 1e-10])
nocc = mol.nelectron // 2
enuc = mol.energy_nuc()
max_cycle = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def run(name, damp_its=0, damp=0.5, shift_its=0, shift=0.5, diis_start=1, diis_space=8, dm0=None):
    mf = jp.apply(RKS(mol, h, S, Grids(mol, 30, 8), xc="b3lyp", numint=ClosedFormNumInt()), jp.get_default_config())
    _, c = np.linalg.eigh(X.T @ h @ X)
    c = X @ c
    dm = 2.0 * c[:, :nocc] @ c[:, :nocc].T if dm0 is None else dm0
    dm_last, v_last, e_last = 0, 0, 0.0
    errs, focks = [], []
    t0 = time.time()
    for it in range(max_cycle):
        veff = mf.get_veff(mol, dm, dm_last=dm_last, vhf_last=v_last, hermi=1)
        dm_last, v_last = dm, veff
        F = h + _strict(veff)
        e_tot = float(np.einsum("ij,ji->", dm, h)) + float(veff.ecoul) + float(veff.exc) + enuc
        err = X.T @ (F @ dm @ S - S @ dm @ F) @ X
        emax = float(np.abs(err).max())
        print(f"{name} it {it:2d} E {e_tot:.8f} dE {e_tot - e_last:+.2e} err {emax:.2e} t {time.time() - t0:.1f}", flush=True)
        if abs(e_tot - e_last) < 1e-9 and emax < 1e-6:
            print(name, "CONVERGED in", it + 1, flush=True)
            return
        e_last = e_tot
        if it >= diis_start:
            focks.append(F); errs.append(err)
            focks, errs = focks[-diis_space:], errs[-diis_space:]
        if len(errs) > 1:
            n = len(errs)
            B = -np.ones((n + 1, n + 1)); B[n, n] = 0
            for a in range(n):
                for b in range(n):
                    B[a, b] = float(np.vdot(errs[a], errs[b]))
            rhs = np.zeros(n + 1); rhs[n] = -1
            try:
                w = np.linalg.solve(B, rhs)[:n]
                F = sum(wi * Fi for wi, Fi in zip(w, focks))
            except np.linalg.LinAlgError:
                pass
        if it < shift_its:                     # level shift: raise the virtual block by `shift`
            F = F + shift * (S - 0.5 * S @ dm @ S)
        e, cc = np.linalg.eigh(X.T @ F @ X)
        cnew = X @ cc
        dnew = 2.0 * cnew[:, :nocc] @ cnew[:, :nocc].T
        dm = damp * dm + (1 - damp) * dnew if it < damp_its else dnew
    print(name, "NOT converged", flush=True)


t = time.time()
sad = atomic_density_guess(mol)
print("atomic guess", round(time.time() - t, 2), "s; electrons", float(np.trace(sad @ S)), flush=True)
run("atomic_guess_diis", diis_start=0, dm0=sad)
if "--all" in sys.argv:
    run("plain_diis", diis_start=0)
    run("damp8", damp_its=8, damp=0.6, diis_start=8)
    run("shift10", shift_its=10, shift=0.5, diis_start=2)
