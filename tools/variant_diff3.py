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
res = []
for spec in sys.argv[3:]:
    v, kc, ns = spec.split(":")
    os.environ["JQC_JK_ALGO"] = "v%d" % int(v, 0)
    jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = int(kc), int(ns)
    router.gen_jk_kernel.cache_clear()
    g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    vj, vk = g(mol, dm, hermi=1)
    torch.cuda.synchronize()
    c = g.stats["tile_counts"].cpu().numpy()[0].copy()
    res.append(c)
    print(spec, "rows", len(c), "sum", c.sum())
a, b = res[0], res[1]
bad = np.nonzero(a != b)[0]
print("rows differing", len(bad), "of", (a > 0).sum(), "nonzero rows; first:", [(int(i), int(a[i]), int(b[i])) for i in bad[:12]])
# which task rows (nij, nkl) are those
st = g.stats
