"""Mixed-precision J/K (cutoff_fp32 = 1e-13, cutoff_fp64 = 1e-7, the reference's benchmark setting, benchmarks/benchmark_jk.py:120)
vs pure fp64 on a large molecule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.gto import mole
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
name, basis = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
res = {}
for label, c64 in (("fp64", 1e-13), ("mixed", 1e-7)):
    g = jkmod.generate_jk_kernel(lay, cutoff_fp64=c64, cutoff_fp32=1e-13)
    for it in range(3):
        torch.cuda.synchronize(); t = time.time(); vj, vk = g(mol, dm, hermi=1); torch.cuda.synchronize(); dt = time.time() - t
    n64, n32, _ = g.quartet_counts()
    res[label] = (vj, vk)
    print(f"{name}/{basis} {label}: {dt:.3f} s  fp64 quartets {n64:.3e}  fp32 quartets {n32:.3e}", flush=True)
print(f"max|dJ| {float((res['mixed'][0]-res['fp64'][0]).abs().max()):.2e}  max|dK| {float((res['mixed'][1]-res['fp64'][1]).abs().max()):.2e}")
