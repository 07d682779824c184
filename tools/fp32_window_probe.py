import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload("0112-elongated-nitrogenous")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
nocc = mol.nelectron // 2
np.random.seed(9)
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
def run(g):
    g(mol, dm, hermi=1); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(2): out = g(mol, dm, hermi=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 2 * 1e3, out
g64 = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
t64, ref = run(g64)
gm = jkmod.generate_jk_kernel(lay, cutoff_fp64=1e-7, cutoff_fp32=1e-13)
tm, out = run(gm)
n64, n32, per = gm.quartet_counts()
print(f"JQC_FP32_WINDOW={os.environ.get('JQC_FP32_WINDOW')}: fp64 build {t64:.1f} ms, mixed build {tm:.1f} ms, fp32 share {n32 / (n64 + n32):.3f} ({n32:.3e} of {n64 + n32:.3e}), dev J {float((out[0]-ref[0]).abs().max()/ref[0].abs().max()):.2e} K {float((out[1]-ref[1]).abs().max()/ref[1].abs().max()):.2e}")
