"""Wall time of the force kernels: two-electron gradient (jk_grad) and XC gradient (GGA, Becke grid) on a stand-in molecule,
next to one J/K build and one rho + vxc pair of the same inputs.    usage: python tools/grad_bench.py [workload] [basis]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.constants import tile_width
from joltqc_amd.gto import mole
from joltqc_amd.gto.grids import Grids
from joltqc_amd.pyscf import grad, jk as jkmod, rks
from joltqc_amd.pyscf.basis import BasisLayout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "0112-elongated-nitrogenous"
basis = sys.argv[2] if len(sys.argv) > 2 else "def2-svp"
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()


def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
t_jk = timed(lambda: g(mol, dm, hermi=1))
fn = grad.generate_jk_energy_per_atom(lay, cutoff=1e-13)
t_g = timed(lambda: fn(mol, dm), 1)
print(f"{name}/{basis} nao={mol.nao}: J/K build {t_jk:.1f} ms ({g.quartet_counts()[0]:.3e} quartets); two-electron gradient "
      f"{t_g:.1f} ms ({fn.quartet_count():.3e} quartets, {fn.stats['launches']} launches)", flush=True)
lay1 = BasisLayout.from_mol(mol, alignment=1)
gg = Grids(mol, 30, 8).build()
order = rks.arg_group_grids(gg.coords)
n = gg.coords.shape[0] // 256 * 256
class G: pass
gr = G(); gr.coords = gg.coords[order][:n]; gr.weights = gg.weights[order][:n]
rks_fun, rho_k, vxc_k = rks.generate_rks_kernel(lay1)
wv = torch.rand((4, n), dtype=torch.float64, device="cuda") * torch.from_numpy(gr.weights).cuda()
t_rho = timed(lambda: rho_k(mol, gr, "GGA", dm))
t_vxc = timed(lambda: vxc_k(mol, gr, "GGA", wv))
t_xg = timed(lambda: rks_fun.xcgrad_fun(mol, gr, "GGA", dm, wv))
print(f"  grid {n} points: rho {t_rho:.2f} ms, vxc {t_vxc:.2f} ms, XC gradient (GGA) {t_xg:.2f} ms", flush=True)
