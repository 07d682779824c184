"""Compare J/K and quartet counts of ONE class under two kernel variants on a large molecule.
usage: python tools/variant_diff.py <class> <variantA> <variantB> [workload]"""
import os, sys
cls, va, vb = sys.argv[1], int(sys.argv[2], 0), int(sys.argv[3], 0)
os.environ["JQC_ONLY_CLASS"] = cls
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_workload
from joltqc_amd.backend import jk as router
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload(sys.argv[4] if len(sys.argv) > 4 else "0112-elongated-nitrogenous")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
out = []
for v in (va, vb):
    os.environ["JQC_JK_ALGO"] = "v%d" % v
    router.gen_jk_kernel.cache_clear()
    g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    vj, vk = g(mol, dm, hermi=1)
    n64, _, per = g.quartet_counts()
    out.append((vj.clone(), vk.clone(), n64))
    print(f"variant {v:#x}: quartets {n64}  |J|max {float(vj.abs().max()):.6e} |K|max {float(vk.abs().max()):.6e}")
print(f"dJ {float((out[0][0]-out[1][0]).abs().max()):.3e}  dK {float((out[0][1]-out[1][1]).abs().max()):.3e}")
