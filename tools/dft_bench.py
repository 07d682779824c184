"""Throughput of the grid path (rho / vxc) on a stand-in molecule with a synthetic atom-centred grid.
Reports grid points/s and MFMA FLOP rate (2 m^2 256 per block and GEMM) - feeds DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.gto import mole
from joltqc_amd.pyscf import rks
from joltqc_amd.pyscf.basis import BasisLayout
name, basis, npts = sys.argv[1], sys.argv[2], int(sys.argv[3])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
lay = BasisLayout.from_mol(mol, alignment=1)
rng = np.random.default_rng(0)
at = mol.atom_coords()
per = npts // mol.natm
# radial shells x random directions around every atom, then box-sorted like build_grids does
r = np.abs(rng.normal(0, 1.5, (mol.natm, per, 1))) + 0.05
u = rng.normal(size=(mol.natm, per, 3)); u /= np.linalg.norm(u, axis=-1, keepdims=True)
coords = (at[:, None, :] + r * u).reshape(-1, 3)
coords = coords[rks.arg_group_grids(coords)]
n = coords.shape[0] // 256 * 256
class G: pass
g = G(); g.coords = coords[:n]; g.weights = np.full(n, 1e-3)
_, rho_k, vxc_k = rks.generate_rks_kernel(lay)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
print(f"{name}/{basis}: nao={mol.nao} nao_int={lay.nao} ngrids={n}", flush=True)
for xc, ndim in (("LDA", 1), ("GGA", 4), ("MGGA", 5)):
    wv = torch.rand((ndim, n), dtype=torch.float64, device="cuda")
    for fn, arg, label in ((rho_k, dm, "rho"), (vxc_k, wv, "vxc")):
        fn(mol, g, xc, arg); torch.cuda.synchronize()
        t = time.time(); fn(mol, g, xc, arg); torch.cuda.synchronize(); dt = time.time() - t
        m = (rho_k.stats["nrow_h"] + 15) // 16 * 16
        gemms = (1 if ndim < 5 else 4)
        fl = 2.0 * float((m.astype(float) ** 2).sum()) * 256 * gemms
        print(f"  {label} {xc}: {dt*1e3:8.2f} ms  {n/dt:.3e} pts/s  mean AO rows/block {m.mean():.0f}  MFMA {fl/dt/1e12:.2f} TFLOP/s", flush=True)
