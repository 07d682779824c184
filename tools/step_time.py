"""Wall time of one J/K build (default streams, default launch geometry unless the JQC_* knobs say otherwise) + optional
multi-density-matrix timings.  usage: python tools/step_time.py [workload] [ndm]   (ndm: also time hermi=0, 2-DM and 3-DM calls)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload(sys.argv[1] if len(sys.argv) > 1 else "0112-elongated-nitrogenous")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)


def timeit(d, hermi, n=3, **kw):
    g(mol, d, hermi=hermi, **kw); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        g(mol, d, hermi=hermi, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


knobs = {k: v for k, v in os.environ.items() if k.startswith("JQC_")}
t1 = timeit(dm, 1)
print(f"{name}: J+K hermi=1 one matrix {t1:.1f} ms  quartets {g.quartet_counts()[0]}  knobs {knobs}", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "ndm":
    a = torch.rand_like(dm)                                  # non-symmetric
    t0 = timeit(a, 0, 2)
    d2 = torch.stack([dm, dm.flip(0).flip(1).contiguous()])
    t2 = timeit(d2, 1, 2)
    d3 = torch.stack([dm, dm.flip(0).flip(1).contiguous(), 0.5 * dm])
    t3 = timeit(d3, 1, 2)
    print(f"  hermi=0 (stacks [D, D^T]) {t0:.1f} ms = {t0 / t1:.2f} x;  two matrices {t2:.1f} ms = {t2 / t1:.2f} x;  three matrices {t3:.1f} ms = {t3 / t1:.2f} x", flush=True)
    tj, tk = timeit(dm, 1, 2, with_k=False), timeit(dm, 1, 2, with_j=False)
    print(f"  J only {tj:.1f} ms, K only {tk:.1f} ms", flush=True)
