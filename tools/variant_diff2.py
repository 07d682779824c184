"""Repeat ONE class under several settings on a large molecule and compare J/K + counts against the first run.
usage: python tools/variant_diff2.py <class> <workload> "<variant>:<kchunk_max>:<nsplit_max>" ..."""
import os, sys
cls = sys.argv[1]
os.environ["JQC_ONLY_CLASS"] = cls
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_workload
from joltqc_amd.backend import jk as router
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload(sys.argv[2])
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
ref = None
for spec in sys.argv[3:]:
    v, kc, ns = spec.split(":")
    os.environ["JQC_JK_ALGO"] = "v%d" % int(v, 0)
    jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = int(kc), int(ns)
    router.gen_jk_kernel.cache_clear()
    g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    vj, vk = g(mol, dm, hermi=1)
    n64, _, per = g.quartet_counts()
    if ref is None:
        ref = (vj.clone(), vk.clone())
    st = g.stats["stamps"].cpu().numpy()
    print("   dbg: iterations", int(st[32 - 21]), "with disagreeing nact", int(st[32 - 20]))
    print(f"{spec:>16s}: quartets {n64}  dJ {float((vj-ref[0]).abs().max()):.3e} dK {float((vk-ref[1]).abs().max()):.3e}  |J|max {float(vj.abs().max()):.5e}", flush=True)
