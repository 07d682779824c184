"""Per-class timing of the tiled J/K kernels (serialised on one stream) vs the FLOP model.
usage: JQC_STREAMS=1 python tools/class_profile.py [benzene|<xyz name>]        (JQC_PROFILE_MODE=j | k: J-only / K-only kernels)"""
import os, sys, json
os.environ.setdefault("JQC_STREAMS", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
from joltqc_amd.roofline import quartet_flops

mol, name = load_workload(sys.argv[1] if len(sys.argv) > 1 else "benzene")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
mode = os.environ.get("JQC_PROFILE_MODE", "jk")
kw = {"with_j": "j" in mode, "with_k": "k" in mode}
for _ in range(2): g(mol, dm, hermi=1, **kw)
torch.cuda.synchronize()
g.set_probe("all")
for _ in range(3): g(mol, dm, hermi=1, **kw)
torch.cuda.synchronize()
n64, n32, per = g.quartet_counts()
fl, cnt = {}, {}
for (ang, npr), (a, b) in per.items():
    fl[ang] = fl.get(ang, 0) + a * quartet_flops(ang, npr); cnt[ang] = cnt.get(ang, 0) + a
tm = {}
for ang, (e0, e1) in zip(g.stats["probe_classes"], g.stats["probe_events"]):
    tm.setdefault(ang, []).append(e0.elapsed_time(e1))
rows = []
for ang in tm:
    ms = float(np.min(tm[ang])); rows.append((ms, ang, cnt.get(ang, 0), fl.get(ang, 0)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{name}: total serial kernel time {tot:.2f} ms, quartets {n64}, model GFLOP {sum(fl.values())/1e9:.1f}")
for ms, ang, c, f in rows:
    print(f"  {ang}  {ms:8.3f} ms  quartets {c:9d}  GFLOP {f/1e9:8.3f}  TFLOP/s {f/ms/1e9:7.3f}  Mq/s {c/ms/1e3:8.1f}")
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
json.dump([{"ang": a, "ms": m, "quartets": c, "flop": f} for m, a, c, f in rows], open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "class_profile.json"), "w"))
