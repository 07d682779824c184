"""Full-size cross-check of the scheme table: every class's chosen kernel variant (default launch geometry) against
the plain single-buffered row-lane kernel with one ket pair per workgroup, class by class, on a large molecule.
Role of the reference autotuner's 1q1t == 1qnt assertion (jqc/backend/data/generate_fragment.py:278-309).
usage: python tools/verify_scheme.py [workload] [tolerance]   -> gpurun_out/verify_scheme_<workload>.json
Also run by tests/test_jk_fullsize_gpu.py (``run`` below) with forced ket chunks -- the regime in which the wrong-result
kernel builds of round 1 showed up."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = 0x221          # row-lane, <= 256 VGPRs (no AGPR spill space), Rys table through L2, single TRR buffer


def run(wl="0112-elongated-nitrogenous", tol=1e-10, verbose=True, modes=(("jk", True, True, None),)):
    """Returns (bad, per-class dict).  ``modes``: (name, with_j, with_k, omega) builds to cross-check."""
    import numpy as np, torch
    from bench import load_workload
    from joltqc_amd.backend import jk as router
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    mol, name = load_workload(wl)
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
    angs = sorted(set(int(a) for a in lay.angs))
    classes = [(a, b, c, d) for a in angs for b in angs for c in angs for d in angs if a >= b and a >= c and c >= d]
    out, bad = {}, []
    saved = {k: os.environ.get(k) for k in ("JQC_ONLY_CLASS", "JQC_JK_ALGO")}
    kc, ns = jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX
    try:
        for mode, wj, wk, omega in modes:
            for ang in classes:
                key = "%d%d%d%d" % ang
                os.environ["JQC_ONLY_CLASS"] = key
                os.environ.pop("JQC_JK_ALGO", None)
                if os.environ.get("JQC_VERIFY_ALGO"):          # diagnostics: this variant code instead of the table's choice
                    os.environ["JQC_JK_ALGO"] = "v%d" % int(os.environ["JQC_VERIFY_ALGO"], 0)
                router.gen_jk_kernel.cache_clear()
                jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = kc, ns
                g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
                vj, vk = g(mol, dm, hermi=1, with_j=wj, with_k=wk, omega=omega)
                n1 = g.quartet_counts()[0]
                os.environ["JQC_JK_ALGO"] = "v%d" % REF
                router.gen_jk_kernel.cache_clear()
                jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = 1, 1
                g2 = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
                rj, rk = g2(mol, dm, hermi=1, with_j=wj, with_k=wk, omega=omega)
                n2 = g2.quartet_counts()[0]
                sc = max(float(rj.abs().max()) if wj else 0.0, float(rk.abs().max()) if wk else 0.0, 1e-300)
                ej = float((vj - rj).abs().max()) / sc if wj else 0.0
                ek = float((vk - rk).abs().max()) / sc if wk else 0.0
                ok = ej < tol and ek < tol and n1 == n2
                out[mode + ":" + key] = {"dJ": ej, "dK": ek, "n": n1, "n_ref": n2, "ok": ok}
                if not ok:
                    bad.append(mode + ":" + key)
                if verbose:
                    print(f"{mode} {key}: dJ {ej:.2e} dK {ek:.2e} quartets {n1} vs {n2} {'ok' if ok else 'MISMATCH'}", flush=True)
    finally:
        jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = kc, ns
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        router.gen_jk_kernel.cache_clear()
    return bad, out


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "0112-elongated-nitrogenous"
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-10
    bad, out = run(wl, tol)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"bad": bad, "classes": out}, open(os.path.join(ROOT, "gpurun_out", f"verify_scheme_{wl}.json"), "w"))
    print("MISMATCHING CLASSES:", bad)
