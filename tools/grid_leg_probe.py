"""Why is the bench's grid leg slower inside bench.py than alone?  grid_leg before and after a J/K call in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = bench.load_workload("0112-elongated-nitrogenous")
r = bench.grid_leg(mol); print("fresh process:", r["rho"]["ms"], r["vxc"]["ms"], flush=True)
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
g(mol, dm, hermi=1); torch.cuda.synchronize()
r = bench.grid_leg(mol); print("after one J/K call:", r["rho"]["ms"], r["vxc"]["ms"], flush=True)
print(torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30)
torch.cuda.empty_cache()
r = bench.grid_leg(mol); print("after empty_cache:", r["rho"]["ms"], r["vxc"]["ms"], flush=True)
