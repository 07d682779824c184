"""Per-class autotuner of the tiled J/K kernels for gfx950 (role of the reference's
jqc/backend/data/generate_fragment.py + optimal_scheme_<GPU>_fp64.json).

  build   (CPU, here)  : compile every candidate variant of every class into joltqc_amd/csrc/kcache_tune
  run     (GPU box)    : time every class under every variant on a workload -> gpurun_out/autotune_<workload>.json
  merge   (CPU)        : best variant per class -> joltqc_amd/data/gfx950_scheme.json ("fp64" table, or "fp32" with JQC_TUNE_FP32=1)
  merge-small (CPU)    : classes whose best variant on a benzene-size workload differs -> the "fp64_small" override table

usage: python tools/autotune.py build | run [workload] | merge <json> [<json2> ...]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = os.path.join(ROOT, "joltqc_amd", "csrc", "kcache_tune")
AVAIL = os.path.join(CACHE, "available.json")
os.environ["JQC_KERNEL_CACHE"] = CACHE
os.environ.setdefault("JQC_TRUST_KERNELS", "1")      # candidates are timed, not used: the adopted table goes through the gates

MINW = lambda n: n << 4
RYS_L2, ST1, WSYNC, CJR = 1 << 8, 1 << 9, 1 << 10, 1 << 11
NKS = lambda log2n: log2n << 12
CANDIDATES = [2 | MINW(1), 2 | MINW(2), 2 | MINW(3),
              2 | MINW(2) | NKS(1), 2 | MINW(2) | NKS(2), 2 | MINW(2) | NKS(3), 2 | MINW(3) | NKS(1),
              1 | MINW(1), 1 | MINW(2), 1 | MINW(1) | RYS_L2, 1 | MINW(2) | RYS_L2, 1 | MINW(2) | RYS_L2 | ST1,
              1 | MINW(3) | RYS_L2 | ST1,
              1 | MINW(1) | WSYNC, 1 | MINW(2) | WSYNC, 1 | MINW(1) | RYS_L2 | WSYNC, 1 | MINW(2) | RYS_L2 | WSYNC,
              1 | MINW(3) | RYS_L2 | WSYNC,
              1 | MINW(1) | CJR, 1 | MINW(1) | RYS_L2 | CJR, 1 | MINW(2) | RYS_L2 | CJR, 1 | MINW(2) | RYS_L2 | ST1 | CJR,
              1 | MINW(3) | RYS_L2 | ST1 | CJR]
# round 3: owner reduction (ORED), per-root phase A (PAROOT) and integral-chunk caps for the row-lane forms.  JQC_TUNE_SET=ored:
# the row-lane entries of the current table (baseline) + the new forms, for the classes of JQC_TUNE_CLASSES
ORED, PAROOT = 1 << 18, 1 << 19
ECAP = lambda code: code << 16
ORED_SET = sorted({0x21, 0x121, 0x421, 0x521, 0x921, 0x10421, 0x10521, 0x10321, 0x10921} |
                  {1 | MINW(2) | RYS_L2 | ORED | p | w | c | ECAP(e) for p in (0, PAROOT) for w in (0, WSYNC) for c in (0, CJR)
                   for e in (0, 1, 3)} |
                  {1 | MINW(2) | ORED | p | w | c for p in (0, PAROOT) for w in (0, WSYNC) for c in (0, CJR)})
if os.environ.get("JQC_TUNE_SET") == "ored":
    CANDIDATES = ORED_SET
# round 4: Rys roots in groups through phase A / phase B (RSPLIT: half / a third / a quarter of the TRR array in LDS -> a second
# workgroup per CU where the array was what kept it out).  JQC_TUNE_SET=rsplit: the owner-reduction forms of the table x root groups
RS = lambda code: code << 22
RSPLIT_SET = sorted({b | RS(c) for b in (0xc0d21, 0x40d21, 0xc0521, 0x40521, 0xc0921, 0x40921, 0xc0c21, 0x40c21, 0x40121, 0xc0121)
                     for c in (0, 1, 2, 3)} | {0x921 | RS(1), 0x521 | RS(1), 0x121 | RS(1)})
if os.environ.get("JQC_TUNE_SET") == "rsplit":
    CANDIDATES = RSPLIT_SET
if os.environ.get("JQC_TUNE_ONLY"):
    CANDIDATES = [int(x, 0) for x in os.environ["JQC_TUNE_ONLY"].split(",")]
MAX_1Q = 200


def nint(ang):
    n = 1
    for l in ang:
        n *= (l + 1) * (l + 2) // 2
    return n


def classes(lmax=3):
    allc = [(a, b, c, d) for a in range(lmax + 1) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]
    only = os.environ.get("JQC_TUNE_CLASSES")
    if only and lmax == 3:
        keep = {tuple(int(ch) for ch in k) for k in only.split(",")}
        allc = [a for a in allc if a in keep]
    return allc


FP32 = int(os.environ.get("JQC_TUNE_FP32", "0"))     # 1: tune the fp32 kernels (every quartet through them)


def _compile(job):
    from joltqc_amd.backend import lib as L
    ang, v = job
    try:
        L.gen_jk_kernel(ang, 1, 1, 0, FP32, v, compile_only=True)
        return (ang, v, True)
    except Exception:  # noqa: BLE001  (LDS overflow of a variant is an expected outcome)
        return (ang, v, False)


def build():
    from multiprocessing import get_context
    from joltqc_amd.backend import lib as L
    L.lib()
    jobs = [(ang, v) for ang in classes() for v in CANDIDATES if (v & 0xf) != 2 or nint(ang) <= MAX_1Q]
    with get_context("spawn").Pool(8) as pool:
        res = list(pool.imap_unordered(_compile, jobs, chunksize=1))
    ok = sorted(["%d%d%d%d:%d" % (*a, v) for a, v, good in res if good])
    json.dump(ok, open(AVAIL, "w"))
    print(f"{len(ok)} of {len(jobs)} variants built")


def run(workload):
    os.environ["JQC_STREAMS"] = "1"
    import numpy as np, torch
    from bench import load_workload
    from joltqc_amd.backend import jk as router
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    avail = set(json.load(open(AVAIL)))
    mol, name = load_workload(workload)
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
    out = {}
    state = {"v": None}
    default = router.select_algo

    def forced(ang, fp32=False):
        v = state["v"]
        return v if "%d%d%d%d:%d" % (*ang, v) in avail else -1
    for v in CANDIDATES:
        if not allowed(v) and not os.environ.get("JQC_TUNE_ALL"):
            continue          # (one-wave row-lane builds are banned from the table and have faulted on large inputs: not even timed)
        state["v"] = v
        # classes the variant does not exist for are skipped (select -> untimed default kernel)
        router.select_algo = lambda ang, fp32=False, small=False: (forced(ang) if (forced(ang) >= 0 and bool(fp32) == bool(FP32))
                                                                   else default(ang, fp32, small))
        g = jkmod.generate_jk_kernel(lay, 1e100 if FP32 else 1e-13, 1e-13)
        g(mol, dm, hermi=1)
        torch.cuda.synchronize()
        g.set_probe("all")
        for _ in range(int(os.environ.get("JQC_TUNE_REPS", "2"))):
            g(mol, dm, hermi=1)
        torch.cuda.synchronize()
        tm = {}
        for ang, (e0, e1) in zip(g.stats["probe_classes"], g.stats["probe_events"]):
            if forced(ang) >= 0:
                key = "%d%d%d%d" % tuple(ang)
                tm[key] = min(tm.get(key, 1e30), e0.elapsed_time(e1))
        out[str(v)] = tm
        print(f"variant {v:#x}: {len(tm)} classes, sum {sum(tm.values()):.1f} ms", flush=True)
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(out, open(f"gpurun_out/autotune_{workload}{'_fp32' if FP32 else ''}.json", "w"))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/autotune_{workload}{'_fp32' if FP32 else ''}.json", "w"))


def allowed(v):
    """Builds the scheme table may use.  Excluded: one-wave-per-SIMD builds (up to 512 registers per lane: 256 VGPRs + AGPR
    spill space) and three-waves-per-SIMD builds of the row-lane kernels (168 VGPRs, heavy scratch spilling).  Every
    wrong-result kernel build found in round 1 ((gg|fp), the long-range builds of (fp|ff) and (ff|fs)) was one of these,
    the same source being correct at two waves per SIMD; the measured cost of the exclusion is 2.5 %
    (profiles/r01_autotune_last_run.json).  The lane-per-quartet kernels at three waves per SIMD pass every gate."""
    minw = (v >> 4) & 0xf
    if (v & 0xf) == 2:
        # lane-per-quartet kernels: no build of this mode has failed a gate, including the 512-register ones the compiler
        # produces by itself when several ket pairs per iteration (NKS) push LDS past two workgroups per CU
        return minw != 1
    return minw != 1 and not ((v & 0xf) == 1 and minw == 3)


def merge(files):
    """Best variant per class; the first file decides, later files only fill classes the first lacks.  Variants listed
    under "rejected" in the scheme file (they failed tools/verify_scheme.py at full size) are never chosen."""
    path = os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")
    rejected = {k: set(v) for k, v in json.load(open(path)).get("rejected", {}).items()}
    best = {}
    for f in files:
        data = json.load(open(f))
        per = {}
        for v, tm in data.items():
            for key, ms in tm.items():
                if int(v) in rejected.get(key, ()) or not allowed(int(v)):
                    continue
                if key not in per or ms < per[key][1]:
                    per[key] = (int(v), ms)
        for key, (v, ms) in per.items():
            best.setdefault(key, v)
    old = json.load(open(path))
    for prec in (("fp32",) if FP32 else ("fp64",)):          # JQC_TUNE_FP32=1: the files are fp32 timings
        for ang in classes(4):
            key = str(1000 * ang[0] + 100 * ang[1] + 10 * ang[2] + ang[3])
            k4 = "%d%d%d%d" % ang
            if k4 in best:
                old[prec][key] = best[k4]
    old["_comment"] = ("gfx950 (MI355X) scheme table of the J/K kernels, measured per class by tools/autotune.py on a 112-atom "
                       "CHNO molecule / def2-TZVPP. Value = algorithm | variant bits (include/jqc_hip.h): low 4 bits 1 = tiled "
                       "row-lane kernel, 2 = tiled one-quartet-per-lane kernel, 3 = row-lane with 512-thread workgroups; bits 4-7 "
                       "waves per SIMD; bit 8 Rys table through L2; bit 9 single TRR buffer. g classes (not measured) keep rule-based "
                       "entries. Role of the reference's optimal_scheme_<GPU>_fp64.json (jqc/backend/data).")
    json.dump(old, open(path, "w"), indent=0)
    print("wrote", path, {k: hex(v) for k, v in sorted(best.items())})


def merge_small(files):
    """"fp64_small" table: classes whose best variant on a benzene-size workload differs from the main table's and beats
    it there by more than 3 % (launches with fewer workgroups than the chunking target use this table)."""
    path = os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")
    sch = json.load(open(path))
    data = json.load(open(files[0]))
    small = {}
    for ang in classes(3):
        k4, key = "%d%d%d%d" % ang, str(1000 * ang[0] + 100 * ang[1] + 10 * ang[2] + ang[3])
        main = sch["fp64"].get(key)
        eq = {0x221: 0x21, 0x321: 0x121, 0xb21: 0x921}           # bit 9 is a no-op: same build
        t_main = data.get(str(eq.get(main, main)), {}).get(k4)
        cands = [(tm[k4], int(v)) for v, tm in data.items() if k4 in tm and allowed(int(v))]
        if t_main is None or not cands:
            continue
        t, v = min(cands)
        if v != eq.get(main, main) and t < 0.97 * t_main:
            small[key] = v
            print(k4, hex(main), "->", hex(v), "%.1f -> %.1f us" % (1e3 * t_main, 1e3 * t))
    sch["fp64_small"] = small
    json.dump(sch, open(path, "w"), indent=0)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build":
        build()
    elif cmd == "merge-small":
        merge_small(sys.argv[2:])
    elif cmd == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "0112-elongated-nitrogenous")
    else:
        merge(sys.argv[2:])
