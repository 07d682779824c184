"""Mixed precision by tile pairs (pyscf/jk.py build_tile_plan: tile pairs with a bound <= cutoff_fp64 go to the class's FP32 kernel, the
rest to its FP64 kernel; no tile pair staged twice): per class on a large molecule with an SCF-like density, windows 1e-13 / 1e-7,
  fp64   the FP64 kernel alone (all-FP64 windows)        split   JQC_FP32_TILE_SPLIT=1, every class split
HIP events per class (serial stream), then the whole call on the default streams -> gpurun_out/tile_split_bench.json
usage: python tools/tile_split_bench.py [molecule] [basis] [table]      (table: use the scheme table's "fp32_tile_split" instead of all)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from joltqc_amd.constants import tile_width
from joltqc_amd.gto import mole
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout

name = sys.argv[1] if len(sys.argv) > 1 else "0112-elongated-nitrogenous"
basis = sys.argv[2] if len(sys.argv) > 2 else "def2-tzvpp"
use_table = len(sys.argv) > 3 and sys.argv[3] == "table"
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
res, ref, out = {}, None, {}
# third leg: the reference's per-QUARTET split (fp64 launch on (cutoff_fp64, inf), fp32 launch on (cutoff_fp32, cutoff_fp64]: every tile
# pair staged twice), forced on for every class
for label, c64, env, win in (("fp64", 1e-13, None, None), ("split", 1e-7, None if use_table else "1", None if use_table else "0"),
                            ("window", 1e-7, "0", "1")):
    for k, v in (("JQC_FP32_TILE_SPLIT", env), ("JQC_FP32_WINDOW", win)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    g = jkmod.generate_jk_kernel(lay, cutoff_fp64=c64, cutoff_fp32=1e-13)
    for _ in range(2):
        vj, vk = g(mol, dm, hermi=1)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        t = time.perf_counter(); vj, vk = g(mol, dm, hermi=1); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    n64, n32, per = g.quartet_counts()
    out[label] = {"ms": 1e3 * best, "n64": n64, "n32": n32}
    # per class, serial
    g.set_streams(1)
    g.set_probe("all")
    tm = {}
    for rep in range(3):
        g.stats["probe_events"] = []; g.stats["probe_classes"] = []
        g(mol, dm, hermi=1)
        torch.cuda.synchronize()
        cur = {}
        for ang, (e0, e1) in zip(g.stats["probe_classes"], g.stats["probe_events"]):
            k = "%d%d%d%d" % tuple(ang)
            cur[k] = cur.get(k, 0.0) + e0.elapsed_time(e1)
        for k, v in cur.items():
            tm[k] = min(tm.get(k, 1e30), v)
    cnt = {}
    for (ang, npr), (a, b) in g.quartet_counts()[2].items():
        k = "%d%d%d%d" % tuple(ang)
        cnt[k] = [cnt.get(k, [0, 0])[0] + a, cnt.get(k, [0, 0])[1] + b]
    res[label] = {"ms": tm, "counts": cnt}
    if label == "fp64":
        ref = (vj.clone(), vk.clone())
    if label != "fp64":
        out[label]["max_abs_dev_j"] = float((vj - ref[0]).abs().max()); out[label]["max_abs_dev_k"] = float((vk - ref[1]).abs().max())
        out[label]["max_j"], out[label]["max_k"] = float(ref[0].abs().max()), float(ref[1].abs().max())
print(json.dumps(out))
tot64 = tots = 0.0
totw = 0.0
for k in sorted(res["fp64"]["ms"], key=lambda k: -res["fp64"]["ms"][k]):
    a, b, w = res["fp64"]["ms"][k], res["split"]["ms"].get(k, float("nan")), res["window"]["ms"].get(k, float("nan"))
    n, nw = res["split"]["counts"].get(k, [0, 0]), res["window"]["counts"].get(k, [0, 0])
    tot64 += a; tots += b; totw += w
    print(f"{k}: fp64 {a:8.2f} ms  tile split {b:8.2f} ms ({b / a:.3f}, fp32 share {n[1] / max(n[0] + n[1], 1):.2f})  "
          f"quartet window {w:8.2f} ms ({w / a:.3f}, fp32 share {nw[1] / max(nw[0] + nw[1], 1):.2f})")
print(f"serial sums: fp64 {tot64:.1f} ms  tile split {tots:.1f} ms ({tots / tot64:.3f})  quartet window {totw:.1f} ms ({totw / tot64:.3f})")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"molecule": name, "basis": basis, "whole": out, "per_class": res}, open(os.path.join(ROOT, "gpurun_out", "tile_split_bench.json"), "w"), indent=1)
