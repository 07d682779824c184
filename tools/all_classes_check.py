"""Every angular class (s..g) on a three-atom molecule with an artificial s/p/d/f/g basis, CLASS BY CLASS, against the CPU
oracle restricted to the quartets of that class.  usage: python tools/all_classes_check.py [variant code]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.gto import mole
from joltqc_amd.backend import jk as router
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
from oracle import dense
if len(sys.argv) > 1:
    os.environ["JQC_JK_ALGO"] = "v%d" % int(sys.argv[1], 0)
shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
          [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
mol = mole.Mole(atom="C 0 0 0; C 0 0.3 2.4; H 1.5 0.2 0.9", basis={"C": shells, "H": shells}, unit="B")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = dm @ dm.T
allq = dense.canonical_quartets(lay)
ang_of = np.asarray(lay.angs)
qa = ang_of[allq.astype(int)]                      # [nq, 4] angular momenta
bad = []
for li in range(5):
    for lj in range(li + 1):
        for lk in range(li + 1):
            for ll in range(lk + 1):
                key = "%d%d%d%d" % (li, lj, lk, ll)
                if os.environ.get("JQC_CHECK_ONLY") and key not in os.environ["JQC_CHECK_ONLY"].split(","):
                    continue
                sel = (qa == np.array([li, lj, lk, ll])).all(1)
                if not sel.any():
                    continue
                om = 0.3 if os.environ.get("JQC_CHECK_MODE") == "lr" else None
                rj, rk = dense.get_jk(lay, dm, hermi=1, quartets=allq[sel], omega=om)
                os.environ["JQC_ONLY_CLASS"] = key
                g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
                errs = []
                for rep in range(2):
                    vj, vk = g(mol, dm, hermi=1, omega=om)
                    sc = max(np.abs(rj).max(), np.abs(rk).max(), 1e-300)
                    errs.append(max(np.abs(vj.cpu().numpy() - rj).max(), np.abs(vk.cpu().numpy() - rk).max()) / sc)
                n = g.quartet_counts()[0]
                ok = max(errs) < 1e-10 and n == int(sel.sum())
                if not ok:
                    bad.append(key)
                print(f"{key}: err {errs[0]:.1e} {errs[1]:.1e} quartets {n} of {int(sel.sum())} variant {router.select_algo((li, lj, lk, ll)):#x} {'ok' if ok else 'MISMATCH'}", flush=True)
print("MISMATCHING:", bad)
