"""Fused mixed-precision builds (JQC_VARIANT_MIXED: FP64 phase + packed-FP32 phase in one launch) of the lane-per-quartet classes:
  check   every such class against the CPU oracle on C2H with an artificial s..f basis, in three settings: both windows populated
          (cutoff_fp64 = 1e-4), every quartet through the packed-FP32 phase (cutoff_fp64 = 1e20), all-FP64 baseline;
  time    per class on a large molecule with an SCF-like density (the windows of apply(): 1e-13 / 1e-7): the fp64 kernel alone
          against the fused build, events around the class's launch, best of 3 -> gpurun_out/mixed_class_bench.json
usage: python tools/mixed_class_bench.py check|time|both [molecule] [basis]     (JQC_KERNEL_SRC selects the kernel sources)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("JQC_TRUST_KERNELS", "1")
os.environ.setdefault("JQC_STREAMS", "1")
import numpy as np, torch
from joltqc_amd.backend import jk as router
from joltqc_amd.constants import tile_width
from joltqc_amd.gto import mole
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout

what = sys.argv[1] if len(sys.argv) > 1 else "both"
ALL = [(a, b, c, d) for a in range(4) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]
T1Q = [a for a in ALL if (router.select_algo(a) & 0xf) == 2 or (router.select_algo(a, small=True) & 0xf) == 2]
only = os.environ.get("MIXED_CLASSES")
if only:
    T1Q = [tuple(int(ch) for ch in k) for k in only.split(",")]


def fused(on):
    os.environ["JQC_MIXED_FUSED"] = "1" if on else "0"


def check():
    from oracle import dense
    shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
              [2, [0.8, 1.0]], [3, [0.9, 1.0]]]
    mol = mole.Mole(atom="C 0 0 0; C 0 0.3 2.4; H 1.5 0.2 0.9; C 6.5 0.1 -0.4", basis={"C": shells, "H": shells}, unit="B")
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao); dm = dm @ dm.T
    allq = dense.canonical_quartets(lay)
    qa = np.asarray(lay.angs)[allq.astype(int)]
    worst = {}
    for target in (4096, 1):                     # small-launch table, then main table with long ket chunks
        jkmod.TARGET_WGS = target
        for ang in T1Q:
            sel = (qa == np.array(ang)).all(1)
            if not sel.any():
                continue
            rj, rk = dense.get_jk(lay, dm, hermi=1, quartets=allq[sel])
            sc = max(np.abs(rj).max(), np.abs(rk).max())
            os.environ["JQC_ONLY_CLASS"] = "%d%d%d%d" % ang
            for label, c64, tol in (("both", 1e-4, 3e-9), ("all32", 1e20, 3e-5), ("fp64", 1e-13, 1e-11)):
                fused(label != "fp64")
                g = jkmod.generate_jk_kernel(lay, cutoff_fp64=c64, cutoff_fp32=1e-13)
                vj, vk = g(mol, dm, hermi=1)
                n64, n32, _ = g.quartet_counts()
                err = max(np.abs(vj.cpu().numpy() - rj).max(), np.abs(vk.cpu().numpy() - rk).max()) / sc
                worst[label] = max(worst.get(label, 0.0), err)
                flag = "" if err < tol and n64 + n32 == int(sel.sum()) and (label == "fp64" or n32 > 0) else "   <-- FAIL"
                print(f"target {target:5d} class {ang} {label:6s} rel err {err:.2e}  fp64 {n64} fp32 {n32} of {int(sel.sum())}{flag}", flush=True)
    os.environ.pop("JQC_ONLY_CLASS", None)
    print("worst:", worst, flush=True)
    return worst


def timing(name, basis):
    mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    nocc = mol.nelectron // 2
    c = np.random.rand(mol.nao, nocc) - 0.5
    dm = torch.from_numpy(c @ c.T / nocc).cuda()
    present = set()
    for a in lay.angs:
        present.add(int(a))
    res = {}
    ref = {}
    for label, on, c64 in (("fp64", False, 1e-13), ("fused", True, 1e-7)):
        fused(on)
        g = jkmod.generate_jk_kernel(lay, cutoff_fp64=c64, cutoff_fp32=1e-13)
        g.set_probe("all")
        for ang in T1Q:
            if any(l not in present for l in ang):
                continue
            os.environ["JQC_ONLY_CLASS"] = "%d%d%d%d" % ang
            best = 1e30
            for rep in range(4):
                g.stats["probe_events"] = []; g.stats["probe_classes"] = []
                vj, vk = g(mol, dm, hermi=1)
                torch.cuda.synchronize()
                ms = sum(e0.elapsed_time(e1) for e0, e1 in g.stats["probe_events"])
                if rep:
                    best = min(best, ms)
            n64, n32, _ = g.quartet_counts()
            key = "%d%d%d%d" % ang
            res.setdefault(key, {})[label] = {"ms": best, "n64": n64, "n32": n32}
            if label == "fp64":
                ref[key] = (vj.clone(), vk.clone())
            else:
                sc = float(max(ref[key][0].abs().max(), ref[key][1].abs().max()))
                res[key]["max_abs_dev"] = float(max((vj - ref[key][0]).abs().max(), (vk - ref[key][1]).abs().max()))
                res[key]["scale"] = sc
                r = res[key]
                print(f"{key}: fp64 {r['fp64']['ms']:8.2f} ms  fused {best:8.2f} ms  ratio {best / r['fp64']['ms']:.3f}  fp32 share "
                      f"{n32 / max(n64 + n32, 1):.2f}  max|dev| {r['max_abs_dev']:.1e} (largest element {sc:.1e})", flush=True)
    os.environ.pop("JQC_ONLY_CLASS", None)
    t64 = sum(r["fp64"]["ms"] for r in res.values() if "fused" in r)
    tmx = sum(r["fused"]["ms"] for r in res.values() if "fused" in r)
    print(f"sum over {len(res)} classes: fp64 {t64:.1f} ms, fused {tmx:.1f} ms, ratio {tmx / t64:.3f}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mixed_class_bench.json"), "w") as f:
        json.dump({"molecule": name, "basis": basis, "classes": res}, f, indent=1)


if what in ("check", "both"):
    check()
if what in ("time", "both"):
    timing(sys.argv[2] if len(sys.argv) > 2 else "0112-elongated-nitrogenous", sys.argv[3] if len(sys.argv) > 3 else "def2-tzvpp")
