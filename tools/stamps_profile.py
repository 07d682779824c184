"""Where a workgroup of one class kernel spends its cycles (wave 0 s_memtime stamps, -DSTAMPS=1 diagnostic build).
usage: python tools/stamps_profile.py <class e.g. 2110> [workload]   (GPU box; JITs the stamped kernel into /tmp)"""
import os, sys
cls = sys.argv[1]
os.environ["JQC_ONLY_CLASS"] = cls
os.environ["JQC_EXTRA_DEFS"] = os.environ.get("JQC_EXTRA_DEFS", "") + " -DSTAMPS=1"
os.environ["JQC_KERNEL_CACHE"] = "/tmp/kc_stamps"
os.environ["JQC_TRUST_KERNELS"] = "1"
os.environ.setdefault("JQC_STREAMS", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload(sys.argv[2] if len(sys.argv) > 2 else "benzene")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
for _ in range(3): g(mol, dm, hermi=1)
torch.cuda.synchronize()
st = g.stats["stamps"].cpu().numpy()[::-1].copy()     # counter[-1-k] -> st[k]
nwg = max(int(st[15]), 1)
names = ["lookup", "bra stage", "Dij stage", "ket loads issue", "ket stage stores", "barrier1", "compute", "barrier2", "flush", "Jij flush+exit"]
n64, _, per = g.quartet_counts()
print(f"{name} class {cls}: workgroups {nwg}, quartets {n64}, quartets/WG {n64 / nwg:.1f}")
tot = sum(int(x) for x in st[:15])
from joltqc_amd.backend import jk as _router
tile1q = (_router.select_algo(tuple(int(c) for c in cls)) & 0xf) == 2
lanes = 0
if tile1q and int(st[11]):
    # lane-per-quartet mode: slots 10-12 = batch overhead / integral evaluation / contraction + atomics of wave 0, slot 13 = active lanes
    names += ["  (batch head)", "  (integrals)", "  (contraction)"]
    lanes = int(st[13]); st[13] = 0
    st[6] -= st[10] + st[11] + st[12]
elif int(st[10]) + int(st[11]):
    names += ["  (phase A)", "  (phase B)", "  (step barrier)", "  (owner red.+step head)" if int(st[14]) else "  (contraction)", "  (contraction arith.)"]
    st[6] -= st[10] + st[11] + st[12] + st[13] + st[14]      # compute = remainder outside the stamped inner phases
for k, nm in enumerate(names):
    print(f"  {nm:16s} {int(st[k]) / nwg:10.0f} cycles/WG  {100.0 * int(st[k]) / max(tot, 1):5.1f} %")
print(f"  total            {tot / nwg:10.0f} cycles/WG")
if lanes:
    print(f"  wave 0: {lanes / nwg:.1f} lane-quartets per WG (its share of {n64 / nwg:.1f} quartets/WG over 4 waves)")
