"""Host-side cost of one get_jk call vs. its GPU time (benzene / def2-TZVPP)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload(sys.argv[1] if len(sys.argv) > 1 else "benzene")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
for _ in range(3): g(mol, dm, hermi=1)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
hs = []
for _ in range(n):
    t1 = time.perf_counter(); g(mol, dm, hermi=1); hs.append(time.perf_counter() - t1)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print(f"{name}: host {1e3*th/n:.2f} ms/call (median {1e3*np.median(hs):.2f}), total {1e3*tt/n:.2f} ms/call, launches {g.stats['launches']}")
# one synchronous call
torch.cuda.synchronize(); t1 = time.perf_counter(); g(mol, dm, hermi=1); torch.cuda.synchronize(); print(f"single synchronous call {1e3*(time.perf_counter()-t1):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): g(mol, dm, hermi=1)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
