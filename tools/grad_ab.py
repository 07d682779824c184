"""Per-class wall time of the two-electron gradient kernels (jk_grad_<class>), for A/B runs of kernel variants.
usage: [JQC_EXTRA_DEFS=... JQC_GRAD_COOP=0|1] python tools/grad_ab.py <label> <class> [<class> ...] [-- workload basis]
Each class: one warm-up call (compiles the kernel), one timed call restricted to that class (its screening pass included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.constants import tile_width
from joltqc_amd.gto import mole
from joltqc_amd.pyscf import grad
from joltqc_amd.pyscf.basis import BasisLayout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rest = []
if "--" in args:
    rest = args[args.index("--") + 1:]
    args = args[:args.index("--")]
label, classes = args[0], args[1:]
name = rest[0] if rest else "0112-elongated-nitrogenous"
basis = rest[1] if len(rest) > 1 else "def2-tzvpp"
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
fn = grad.generate_jk_energy_per_atom(lay, cutoff=1e-13)
ref = None
for cl in classes:
    ang = tuple(int(x) for x in cl)
    want = lambda a, ang=ang: tuple(a) == ang
    g = fn(mol, dm, _classes=want)
    torch.cuda.synchronize()
    t = time.perf_counter()
    g = fn(mol, dm, _classes=want)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3
    print(f"{label} {cl} {ms:9.1f} ms  {fn.quartet_count():.3e} quartets  |g|={float(g.abs().sum()):.12e}", flush=True)
