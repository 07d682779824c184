"""J/K step with a spatially DECAYING density, D_ab = exp(-gamma |R_a - R_b|) x random sign/size, next to the dense one: how much of
the density screening's saving in quartets turns into time (tile pairs whose quartets all fail the density test are still
staged).    usage: python tools/decay_density_bench.py [gamma ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload("0112-elongated-nitrogenous")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
rng = np.random.default_rng(9)
# centre of every AO of the molecule's basis
at = mol.atom_coords()
ao_atom = np.concatenate([[int(b[0])] * ((2 * int(b[1]) + 1) * int(b[3])) for b in np.asarray(mol._bas)])
R = at[ao_atom]
dist = np.linalg.norm(R[:, None, :] - R[None, :, :], axis=2)
base = rng.random((mol.nao, mol.nao)) - 0.5
base = base + base.T
for gamma in [0.0] + [float(x) for x in sys.argv[1:]]:
    dm = torch.from_numpy(base * np.exp(-gamma * dist)).cuda()
    g(mol, dm, hermi=1); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2):
        g(mol, dm, hermi=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    n = sum(g.quartet_counts()[:2])
    print(f"gamma {gamma:4.2f}: {dt*1e3:8.1f} ms  quartets {n:.3e}  {n/dt/1e9:.2f} Gq/s", flush=True)
