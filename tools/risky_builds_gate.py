"""Quarantine gate for the kernel builds whose register allocation matches the one family of wrong-result builds ever seen here
(more than 256 registers per lane AND SGPRs spilled into VGPR lanes; profiles/r03_diag_wrong_result_builds.txt).
Every such build of the ahead-of-time set is run TWICE on the benzene molecule with the artificial s..g basis under forced ket chunks
(one workgroup walks many ket tile pairs: the regime of every failure found so far) and must
  * reproduce itself from run to run to 1e-12 of the largest element (the failures were timing dependent), and
  * agree to 1e-10 with the plain row-lane reference variant (<= 256 registers, one ket pair per workgroup).
Result: gpurun_out/risky_builds_gate.json, copied to joltqc_amd/data/ and read by tools/make_manifest.py -- a risky build without a
passing record stays OUT of the verified manifest (it is then cross-checked on first use by pyscf/jk.py).
usage: python tools/risky_builds_gate.py list        (CPU: which builds are risky)
       python tools/risky_builds_gate.py run         (GPU box)"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
REF = 0x221
KEY_RE = re.compile(r"^jk(\d+)_(\d)(\d)(\d)(\d)_j(\d)k(\d)_lr(\d)_(f32|f64)_t")


def resources(path):
    """(vgpr_count, agpr_count, sgpr_spill_count, scratch bytes) of a code object (its AMDGPU metadata note)."""
    out = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, out).group(1))
    return g("vgpr_count"), g("agpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")


def is_risky(vgpr, sspill):
    """More than 256 registers per lane (the upper half only reachable through v_accvgpr copies) AND SGPRs spilled to VGPR lanes."""
    return vgpr > 256 and sspill > 0


def parse_key(key):
    m = KEY_RE.match(key)
    algo, li, lj, lk, ll, dj, dk, lr = (int(x) for x in m.groups()[:8])
    return algo, (li, lj, lk, ll), dj, dk, lr, m.group(9) == "f32"


def aot_builds():
    """{kernel key: code-object path} of the gated ahead-of-time set (what make_manifest.py lists)."""
    import __graft_entry__ as G
    from joltqc_amd.backend import jk as router, lib as L
    tag = L.lib().jqc_source_tag().decode()
    out = {}
    for ang, dj, dk, lr, fp32, algo in G.kernel_jobs():
        if (algo & 0xf) == L.ALGO_1Q1T:
            continue
        router.gen_jk_kernel(ang, bool(dj), bool(dk), bool(lr), bool(fp32), algo, True)
        key = router.kernel_key(ang, dj, dk, lr, fp32, router.resolved_algo(ang, dj, dk, lr, fp32, algo))
        out[key] = os.path.join(L.KERNEL_CACHE, key + "_" + tag + ".hsaco")
    return out


def risky_builds():
    from concurrent.futures import ThreadPoolExecutor
    builds = aot_builds()
    keys = sorted(builds)
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(lambda k: resources(builds[k]), keys))
    return {k: r for k, r in zip(keys, res) if is_risky(r[0], r[2])}, dict(zip(keys, res))


def run():
    import numpy as np, torch
    from bench import load_workload
    from joltqc_amd.backend import jk as router, lib as L
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    os.environ["JQC_TRUST_KERNELS"] = "1"          # (the gate itself is the check; no first-use cross-check on top)
    risky, _ = risky_builds()
    mol, _ = load_workload("benzene-spdfg")
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    d1 = np.random.rand(mol.nao, mol.nao); d1 = d1 @ d1.T
    d2 = np.random.rand(mol.nao, mol.nao); d2 = d2 @ d2.T
    dm1 = torch.from_numpy(d1).cuda()
    dm2 = torch.from_numpy(np.stack([d1, d2])).cuda()
    NDM2 = router.VARIANT_NDM2
    out = {}
    kc, ns, tw = jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX, jkmod.TARGET_WGS

    MIXED = router.VARIANT_MIXED

    def call(algo, ang, dj, dk, lr, fp32, dm, chunks, c64=None):
        os.environ["JQC_ONLY_CLASS"] = "%d%d%d%d" % ang
        os.environ["JQC_JK_ALGO"] = "v%d" % (algo & ~NDM2 & ~MIXED)
        os.environ["JQC_MIXED_FUSED"] = "1" if algo & MIXED else "0"      # (a fused build is reached through the fused launch path)
        router.gen_jk_kernel.cache_clear()
        jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX, jkmod.TARGET_WGS = (kc, ns, 1) if chunks else (1, 1, tw)
        c64 = c64 or (1e100 if fp32 else 1e-13)
        g = jkmod.generate_jk_kernel(lay, cutoff_fp64=c64, cutoff_fp32=1e-13)
        vj, vk = g(mol, dm, hermi=1, with_j=bool(dj), with_k=bool(dk), omega=0.3 if lr else None)
        return [x.clone() for x, on in ((vj, dj), (vk, dk)) if on]

    try:
        for n, key in enumerate(sorted(risky)):
            algo, ang, dj, dk, lr, fp32 = parse_key(key)
            dm = dm2 if algo & NDM2 else dm1
            r = call(REF, ang, dj, dk, lr, False, dm, False)
            sc = max(float(x.abs().max()) for x in r) or 1e-300
            rr = vr = 0.0
            ok = True
            # (a fused mixed-precision build: every quartet through its FP64 phase -- window 1e-13 / 1.0001e-13 --, then every
            #  quartet through its packed-FP32 phase -- window 1e-13 / 1e20)
            for c64, tol in (((1.0001e-13, 1e-10), (1e20, 2e-4)) if algo & MIXED else ((None, 2e-4 if fp32 else 1e-10),)):
                a = call(algo, ang, dj, dk, lr, fp32, dm, True, c64)
                b = call(algo, ang, dj, dk, lr, fp32, dm, True, c64)
                rr = max(rr, max(float((x - y).abs().max()) for x, y in zip(a, b)) / sc)
                v1 = max(float((x - y).abs().max()) for x, y in zip(a, r)) / sc
                vr = max(vr, v1 if tol > 1e-9 else v1 * 1.0)
                ok = ok and v1 <= tol and all(bool(torch.isfinite(x).all()) for x in a)
            ok = ok and rr <= 1e-12
            out[key] = {"run_to_run": rr, "vs_ref": vr, "vgpr": risky[key][0], "sspill": risky[key][2], "ok": bool(ok),
                        "fp32_phase": bool(fp32 or algo & MIXED)}
            if not ok or n % 50 == 0:
                print(f"{n + 1}/{len(risky)} {key}: run-to-run {rr:.1e} vs reference {vr:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    finally:
        jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX, jkmod.TARGET_WGS = kc, ns, tw
        os.environ.pop("JQC_MIXED_FUSED", None)
    rec = {"src_tag": L.lib().jqc_source_tag().decode(), "workload": "benzene, artificial s..g basis, forced ket chunks, each build twice",
           "results": out}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "risky_builds_gate.json"), "w"), indent=0)
    bad = [k for k, v in out.items() if not v["ok"]]
    print(f"{len(out)} risky builds gated, {len(bad)} failed: {bad}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    else:
        risky, allr = risky_builds()
        print(f"{len(risky)} of {len(allr)} gated builds have > 256 registers and SGPR spills to VGPR lanes")
