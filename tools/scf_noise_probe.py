"""Where does the energy jitter of the incremental SCF come from?  ("next round" item 1a.)

The stand-in Kohn-Sham driver of the tests on 112 atoms / B3LYP / def2-SVP through apply(), DIIS from the atomic-density guess.
Every iteration evaluates the potential THREE ways at the same density matrix:
  inc   the path under test: incremental rho / V_xc / J / K (dm_last, vhf_last handed back as the SCF loop does)
  full  the same closures' configuration, rebuilt from scratch (dm_last = 0, grid caches reset)
  ref   all-FP64 windows, from scratch (only printed when the configuration under test is not already all-FP64)
and prints E and the components (ecoul, exc incl. the exact-exchange part) of each, max |V_inc - V_full|, max |vj|, |vk| differences.
At a few iterations the J/K of that iteration's density CHANGE is also evaluated twice by the tiled kernels (run-to-run order of the
FP64 atomics) and once by the independent queue kernels (JQC_JK_ALGO=1q1t): the sparse small-launch regime of a late SCF iteration.
usage: python tools/scf_noise_probe.py [cycles] [default|fp64|both]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import joltqc_amd.pyscf as jp
from joltqc_amd.gto import mole
from joltqc_amd.gto.grids import Grids
from joltqc_amd.pyscf import int1e
from joltqc_amd.pyscf.basis import BasisLayout
from standin_scf import RKS, ClosedFormNumInt, _strict, atomic_density_guess

basis = os.environ.get("PROBE_BASIS", "def2-svp")
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules/0112-elongated-nitrogenous.xyz")), basis=basis)
S, T, V = (x.cpu().numpy() for x in int1e.int1e(BasisLayout.from_mol(mol, alignment=1), mol))
h = T + V
s, U = np.linalg.eigh(S)
X = U[:, s > 1e-10] / np.sqrt(s[s > 1e-10])
nocc = mol.nelectron // 2
enuc = mol.energy_nuc()
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 26
which = sys.argv[2] if len(sys.argv) > 2 else "both"
FP64 = {"jk": {"cutoff_fp32": 1e-13, "cutoff_fp64": 1e-13}, "dft": {"cutoff_fp32": 1e-13, "cutoff_fp64": 1e-13}}
JK_AT = (6, 10, 14, 18, 22)


def make(cfg):
    c = jp.get_default_config()
    if cfg:
        c.update(cfg)
    return jp.apply(RKS(mol, h, S, Grids(mol, 30, 8), xc="b3lyp", numint=ClosedFormNumInt()), c)


def energy(dm, veff):
    return float(np.einsum("ij,ji->", dm, h)) + float(veff.ecoul) + float(veff.exc) + enuc


def scratch(mf, dm):
    mf._numint.nr_rks.__func__.reset_cache()
    return mf.get_veff(mol, dm, dm_last=0, vhf_last=0, hermi=1)


def queue_jk(lay):
    from joltqc_amd.backend import jk as router
    from joltqc_amd.pyscf import jk as jkmod
    os.environ["JQC_JK_ALGO"] = "1q1t"
    router.gen_jk_kernel.cache_clear()
    try:
        return jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    finally:
        del os.environ["JQC_JK_ALGO"]


def run(label, cfg, dm0):
    mf, mf_full = make(cfg), make(cfg)
    mf_ref = make(FP64) if cfg is None else None
    dm, dm_last, v_last, e_last = dm0, 0, 0, 0.0
    errs, focks = [], []
    t0 = time.time()
    for it in range(cycles):
        veff = mf.get_veff(mol, dm, dm_last=dm_last, vhf_last=v_last, hermi=1)
        e_inc = energy(dm, veff)
        vf = scratch(mf_full, dm)
        e_full = energy(dm, vf)
        line = (f"{label} it {it:2d} E_inc {e_inc:.9f} dE {e_inc - e_last:+.2e} | inc-full: E {e_inc - e_full:+.2e} ecoul "
                f"{veff.ecoul - vf.ecoul:+.2e} exc {veff.exc - vf.exc:+.2e} maxV {np.abs(_strict(veff) - _strict(vf)).max():.1e} "
                f"vj {np.abs(veff.vj - vf.vj).max():.1e} vk {np.abs(veff.vk - vf.vk).max():.1e}")
        if mf_ref is not None:
            vr = scratch(mf_ref, dm)
            line += f" | full-ref(fp64): E {e_full - energy(dm, vr):+.2e} exc {vf.exc - vr.exc:+.2e} maxV {np.abs(_strict(vf) - _strict(vr)).max():.1e}"
        F = h + _strict(veff)
        err = X.T @ (F @ dm @ S - S @ dm @ F) @ X
        line += f" | diis_err {np.abs(err).max():.2e} |dD|max {np.abs(dm - dm_last).max():.1e} t {time.time() - t0:.0f}"
        print(line, flush=True)
        if it in JK_AT and it > 0:
            dd = torch.from_numpy(dm - dm_last).cuda()
            g = mf.get_jk
            a = [x.clone() for x in g(mol, dd, hermi=1)]
            b = [x.clone() for x in g(mol, dd, hermi=1)]
            gq = queue_jk(mf._jqc_basis_layout)
            q = gq(mol, dd, hermi=1)
            from joltqc_amd.backend import jk as router
            router.gen_jk_kernel.cache_clear()
            sc = [float(x.abs().max()) for x in a]
            print(f"   J/K of dD at it {it}: |J|max {sc[0]:.2e} |K|max {sc[1]:.2e}; tiled run-to-run dJ {float((a[0]-b[0]).abs().max()):.1e} "
                  f"dK {float((a[1]-b[1]).abs().max()):.1e}; tiled-queue dJ {float((a[0]-q[0]).abs().max()):.1e} dK "
                  f"{float((a[1]-q[1]).abs().max()):.1e}; quartets tiled {g.quartet_counts()[0]:.3e} queue {gq.quartet_counts()[0]:.3e}",
                  flush=True)
        dm_last, v_last, e_last = dm, veff, e_inc
        focks.append(F); errs.append(err)
        focks, errs = focks[-8:], errs[-8:]
        if len(errs) > 1:
            n = len(errs)
            B = -np.ones((n + 1, n + 1)); B[n, n] = 0
            for a_ in range(n):
                for b_ in range(n):
                    B[a_, b_] = float(np.vdot(errs[a_], errs[b_]))
            rhs = np.zeros(n + 1); rhs[n] = -1
            try:
                w = np.linalg.solve(B, rhs)[:n]
                F = sum(wi * Fi for wi, Fi in zip(w, focks))
            except np.linalg.LinAlgError:
                pass
        e, cc = np.linalg.eigh(X.T @ F @ X)
        c = X @ cc
        dm = 2.0 * c[:, :nocc] @ c[:, :nocc].T


sad = atomic_density_guess(mol)
if which in ("default", "both"):
    run("default", None, sad)
if which in ("fp64", "both"):
    run("fp64", FP64, sad)
