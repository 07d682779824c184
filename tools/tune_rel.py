"""Relative re-tuning of the gfx950 scheme table: every class's CURRENT variant with extra variant bits OR-ed in
(e.g. the row-lane integral-chunk cap JQC_VARIANT_ECAP: 0x10000 / 0x20000 / 0x30000), timed per class against the
current table on a large molecule.  Complements tools/autotune.py (which scans absolute variant codes).

  python tools/tune_rel.py build 0x10000,0x20000,0x30000        (CPU: compile into joltqc_amd/csrc/kcache_tune)
  python tools/tune_rel.py run   0x10000,0x20000,0x30000 [workload]   (GPU box) -> gpurun_out/tune_rel_<workload>.json
  python tools/tune_rel.py merge gpurun_out/tune_rel_<workload>.json  (CPU: adopt bits that win by > 2 % and have no scratch
                                                                        penalty; writes gfx950_scheme.json "fp64")
Only classes whose current variant is a row-lane kernel (algorithm 1) take row-lane bits; lane-per-quartet bits (0xc000)
go to algorithm 2 classes.
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = os.path.join(ROOT, "joltqc_amd", "csrc", "kcache_tune")
os.environ["JQC_KERNEL_CACHE"] = CACHE
os.environ["JQC_TRUST_KERNELS"] = "1"
SCHEME = os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")


def classes(lmax=3):
    return [(a, b, c, d) for a in range(lmax + 1) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]


def applies(v, bits):
    algo = v & 0xf
    if bits & 0x30000:
        return algo == 1
    if bits & 0xc000:
        return algo == 2
    return True


def _compile(job):
    from joltqc_amd.backend import jk as router
    ang, v = job
    try:
        router.gen_jk_kernel(ang, True, True, False, False, v, True)
        return None
    except Exception as e:  # noqa: BLE001
        return f"{ang} {v:#x}: {str(e)[-200:]}"


def build(bits_list):
    from multiprocessing import get_context
    from joltqc_amd.backend import jk as router
    jobs = set()
    for ang in classes():
        v = router.select_algo(ang)
        jobs.add((ang, v))
        for b in bits_list:
            if applies(v, b):
                jobs.add((ang, v | b))
    with get_context("spawn").Pool(8) as pool:
        errs = [e for e in pool.imap_unordered(_compile, sorted(jobs), chunksize=1) if e]
    print(len(jobs), "builds,", len(errs), "failed:", errs[:5])


def run(bits_list, workload):
    os.environ["JQC_STREAMS"] = "1"
    import numpy as np, torch
    from bench import load_workload
    from joltqc_amd.backend import jk as router
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    mol, name = load_workload(workload)
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
    default = router.select_algo
    out = {}
    for b in [0] + list(bits_list):
        def sel(ang, fp32=False, small=False, b=b):
            v = default(ang, fp32, small)
            return v | b if (b and not fp32 and applies(v, b)) else v
        router.select_algo = sel
        router.gen_jk_kernel.cache_clear()
        g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
        g(mol, dm, hermi=1)
        torch.cuda.synchronize()
        g.set_probe("all")
        for _ in range(int(os.environ.get("JQC_TUNE_REPS", "2"))):
            g(mol, dm, hermi=1)
        torch.cuda.synchronize()
        tm = {}
        for ang, (e0, e1) in zip(g.stats["probe_classes"], g.stats["probe_events"]):
            if b == 0 or applies(default(ang), b):
                key = "%d%d%d%d" % tuple(ang)
                tm[key] = min(tm.get(key, 1e30), e0.elapsed_time(e1))
        out[hex(b)] = tm
        print(f"bits {b:#x}: {len(tm)} classes, sum {sum(tm.values()):.1f} ms", flush=True)
    router.select_algo = default
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"tune_rel_{workload}.json"), "w"), indent=1)
    base = out["0x0"]
    for k in sorted(base, key=lambda k: -base[k])[:40]:
        print(k, f"{base[k]:8.2f}", "  ".join(f"{b} {out[b].get(k, float('nan')):8.2f}" for b in out if b != "0x0"))


def merge(path):
    data = json.load(open(path))
    sch = json.load(open(SCHEME))
    base = data["0x0"]
    changed = {}
    for k, t0 in base.items():
        key = str(int(k))                     # "0000" -> "0", "2110" -> "2110"
        best_b, best_t = 0, t0
        for b, tm in data.items():
            if b != "0x0" and k in tm and tm[k] < 0.98 * best_t:
                best_b, best_t = int(b, 16), tm[k]
        if best_b:
            sch["fp64"][key] = int(sch["fp64"][key]) | best_b
            changed[k] = (hex(best_b), round(t0, 2), round(best_t, 2))
    json.dump(sch, open(SCHEME, "w"), indent=0)
    print(len(changed), "classes changed:", changed)
    print("sum before %.1f ms, after %.1f ms" % (sum(base.values()), sum(base.values()) - sum(c[1] - c[2] for c in changed.values())))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "merge":
        merge(sys.argv[2])
    else:
        bits = [int(x, 0) for x in sys.argv[2].split(",")]
        if cmd == "build":
            build(bits)
        else:
            run(bits, sys.argv[3] if len(sys.argv) > 3 else "0112-elongated-nitrogenous")
