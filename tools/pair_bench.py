"""J-only build: tiled kernels vs the pair-based backend (pair_vj) on a large molecule.
usage: python tools/pair_bench.py <xyz name> <basis>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from joltqc_amd.constants import tile_width
from joltqc_amd.gto import mole
from joltqc_amd.pyscf import jk as jkmod, jk_pair
from joltqc_amd.pyscf.basis import BasisLayout
name, basis = sys.argv[1], sys.argv[2]
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
tile = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
pair = jk_pair.generate_jk_kernel(lay, 1e-13, 1e-13)
res = {}
for label, fn in (("tiled J-only kernels", tile), ("pair-based J (pair_vj)", pair)):
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        vj, _ = fn(mol, dm, hermi=1, with_k=False)
        torch.cuda.synchronize(); dt = time.time() - t
    res[label] = (dt, vj)
    extra = f" pair classes {pair.stats.get('pair_classes')} tile classes {pair.stats.get('tile_classes')} launches {pair.stats.get('pair_launches')} pair quartet evaluations {int(pair.stats['pair_counter'].item()):.3e}" if fn is pair else f" quartets {tile.quartet_counts()[0]:.3e}"
    print(f"{name}/{basis} {label}: {dt*1e3:9.1f} ms{extra}", flush=True)
a, b = res["tiled J-only kernels"][1], res["pair-based J (pair_vj)"][1]
print(f"max |J_pair - J_tile| / max|J| = {float((a - b).abs().max() / a.abs().max()):.2e}")
