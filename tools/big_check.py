"""Large-molecule J/K run (stand-in geometries from joltqc_amd/data/molecules): timing + quartet counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.gto import mole
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
from joltqc_amd.roofline import quartet_flops
name, basis = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
t0 = time.time(); lay = BasisLayout.from_mol(mol, alignment=tile_width)
print(f"{name}/{basis}: natm={mol.natm} nao={mol.nao} nbas(padded)={lay.nbasis} nao_int={lay.nao} layout {time.time()-t0:.2f}s", flush=True)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
for it in range(3):
    t = time.time(); vj, vk = g(mol, dm, hermi=1); torch.cuda.synchronize(); dt = time.time() - t
    print(f"  call {it}: {dt:.3f}s host {g.stats['host_seconds']:.3f}s launches {g.stats['launches']}", flush=True)
n64, n32, per = g.quartet_counts()
fl = sum(a * quartet_flops(ang, npr) for (ang, npr), (a, b) in per.items())
print(f"  quartets {n64:.4e}  {n64/dt:.3e} q/s  model {fl/1e12:.3f} TFLOP -> {fl/dt/1e12:.2f} TFLOP/s  |vj|max {float(vj.abs().max()):.3e} finite {bool(torch.isfinite(vj).all() and torch.isfinite(vk).all())}")
