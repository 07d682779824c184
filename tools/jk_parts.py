"""J+K vs J-only vs K-only wall time of the tiled kernels on one workload (decides whether J should go to the pair backend when K
is wanted as well).    usage: python tools/jk_parts.py [workload]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload(sys.argv[1] if len(sys.argv) > 1 else "0112-elongated-nitrogenous")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
for label, kw in (("J+K", {}), ("J only", {"with_k": False}), ("K only", {"with_j": False})):
    g(mol, dm, hermi=1, **kw); torch.cuda.synchronize()
    t = time.perf_counter(); g(mol, dm, hermi=1, **kw); torch.cuda.synchronize()
    print(f"{name} {label}: {1e3 * (time.perf_counter() - t):.1f} ms", flush=True)
