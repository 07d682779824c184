import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.gto import mole
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
from oracle import dense
shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
          [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
mol = mole.Mole(atom="C 0 0 0; C 0 0.3 2.4; H 1.5 0.2 0.9", basis={"C": shells, "H": shells}, unit="B")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = dm @ dm.T
allq = dense.canonical_quartets(lay)
qa = np.asarray(lay.angs)[allq.astype(int)]
for key in ("0000", "1000", "2210"):
    ang = np.array([int(c) for c in key]); sel = (qa == ang).all(1)
    os.environ["JQC_ONLY_CLASS"] = key
    for mode, kw, c64 in (("lr", dict(omega=0.3), 1e-13), ("fp32", {}, 1e100)):
        rj, rk = dense.get_jk(lay, dm, hermi=1, quartets=allq[sel], omega=kw.get("omega"))
        g = jkmod.generate_jk_kernel(lay, cutoff_fp64=c64, cutoff_fp32=1e-13)
        vj, vk = g(mol, dm, hermi=1, **kw)
        vj, vk = vj.cpu().numpy(), vk.cpu().numpy()
        print(key, mode, "oracle nan", np.isnan(rj).sum(), np.isnan(rk).sum(), "gpu nan", np.isnan(vj).sum(), np.isnan(vk).sum(),
              "|rj|max", np.nanmax(np.abs(rj)), "|vj|max", np.nanmax(np.abs(vj)), "counts", g.quartet_counts()[:2])
