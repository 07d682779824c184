import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.gto import mole
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
          [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
mol = mole.Mole(atom="C 0 0 0; C 0 0.3 2.4; H 1.5 0.2 0.9", basis={"C": shells, "H": shells}, unit="B")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = dm @ dm.T
b32 = lay.basis_data_fp32["packed"].cpu().numpy()
print("basis32 nan:", np.isnan(b32).sum(), "inf:", np.isinf(b32).sum(), "max", np.nanmax(np.abs(b32)))
for key in ("0000", "1000", "1110", "2110"):
    os.environ["JQC_ONLY_CLASS"] = key
    g = jkmod.generate_jk_kernel(lay, cutoff_fp64=1e100, cutoff_fp32=1e-13)
    res = []
    for rep in range(6):
        vj, vk = g(mol, dm, hermi=1)
        res.append((int(torch.isnan(vj).sum()), int(torch.isnan(vk).sum())))
    tt = g.__closure__ and None
    print(key, res)
import joltqc_amd.pyscf.jk as J
