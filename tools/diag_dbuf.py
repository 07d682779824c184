"""Diagnostics for DESIGN.md 3.1: the double-buffered TRR schedule (TRR_DOUBLE_BUFFER) gives wrong J/K in exactly the builds
that take more than 256 registers per lane ((fd|dd), (fd|fp), (ff|dd), (gd|dd); tools/diag_banned_builds.py).  This tool
rebuilds those classes (and (ff|fp), a > 256-register build that passes) under one perturbation at a time and reports, per
build, the error against the single-buffered two-waves-per-SIMD reference kernel, whether two runs of the same kernel agree
with each other, and whether the error needs more than one ket tile pair per workgroup.
  build (CPU): compile into csrc/kcache_diag from csrc/kernels_dev (the DBUF_DIAG barriers live only in that scratch copy)
  run   (GPU): benzene with the artificial s..g basis
usage: python tools/diag_dbuf.py build|run"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("JQC_DIAG_SET", "dbuf") == "dbuf":
    CLASSES = [(3, 2, 2, 2), (3, 2, 3, 1), (3, 3, 2, 2), (4, 2, 2, 2), (3, 3, 3, 1)]
    DB = "-DTRR_DOUBLE_BUFFER=1"
    VARIANT = 0x111          # row lanes, one wave per SIMD
    PERTURB = [("dbuf", DB), ("dbuf_both_barriers", DB + " -DDBUF_DIAG=3"),
               ("dbuf_sgpr_spills_to_memory", DB + " -mllvm -amdgpu-spill-sgpr-to-vgpr=0"),
               ("dbuf_karg_reload", DB + " -DKARG_RELOAD=1"), ("dbuf_rolled_roots", DB + " -DUNROLL_B=0"),
               ("dbuf_sleep_at_iteration_top", DB + " -DDBUF_DIAG=4"), ("dbuf_barrier_at_iteration_top", DB + " -DDBUF_DIAG=8"),
               ("dbuf_all_waits_zero", DB + " -mllvm -amdgpu-waitcnt-forcezero=1"),
               ("dbuf_O1", DB + " -O1")]
else:   # "wsync": the one-wave-per-SIMD build of the wave-local (gp|ff) kernel, which raises a memory fault
    CLASSES = [(4, 1, 3, 2), (4, 1, 3, 3)]
    VARIANT = 0x511          # row lanes, wave-local steps, one wave per SIMD
    PERTURB = [("wsync_minw1", ""), ("wsync_minw1_sgpr_spills_to_memory", "-mllvm -amdgpu-spill-sgpr-to-vgpr=0"),
               ("wsync_minw1_karg_reload", "-DKARG_RELOAD=1"), ("wsync_minw1_all_waits_zero", "-mllvm -amdgpu-waitcnt-forcezero=1"),
               ("wsync_minw1_rolled_roots", "-DUNROLL_B=0")]
CACHE = os.path.join(ROOT, "joltqc_amd", "csrc", "kcache_diag")
DEV = os.path.join(ROOT, "joltqc_amd", "csrc", "kernels_dev")


def env_for(defs):
    e = dict(os.environ)
    e.update(JQC_KERNEL_SRC=DEV, JQC_KERNEL_CACHE=CACHE, JQC_EXTRA_DEFS=defs, JQC_TRUST_KERNELS="1", JQC_STREAMS="1",
             JQC_TARGET_WGS="32")
    return e


def child_build():
    from joltqc_amd.backend import jk as router
    for ang in CLASSES:
        for algo in (VARIANT, 0x221):
            router.gen_jk_kernel(ang, True, True, False, False, algo, True)
    print("built", repr(os.environ.get("JQC_EXTRA_DEFS")), flush=True)


def child_run(name):
    import numpy as np, torch
    from bench import load_workload
    from joltqc_amd.backend import jk as router
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    mol, _ = load_workload("benzene-spdfg")
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()

    def build(algo, kchunk):
        os.environ["JQC_JK_ALGO"] = "v%d" % algo
        router.gen_jk_kernel.cache_clear()
        jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = kchunk, 1
        g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
        vj, vk = g(mol, dm, hermi=1)
        return vj.clone(), vk.clone()

    for ang in CLASSES:
        key = "%d%d%d%d" % ang
        os.environ["JQC_ONLY_CLASS"] = key
        rj, rk = build(0x221, 1)
        sc = max(float(rj.abs().max()), float(rk.abs().max()))
        line = f"{name:28s} {key}:"
        for kchunk in (1, 8):
            a = build(VARIANT, kchunk)
            b = build(VARIANT, kchunk)
            err = max(float((a[0] - rj).abs().max()), float((a[1] - rk).abs().max())) / sc
            rep = max(float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max())) / sc
            nbad = int(((a[1] - rk).abs() > 1e-9 * sc).sum())
            line += f"  kchunk {kchunk}: err {err:.1e} run-to-run {rep:.1e} wrong K elements {nbad}/{rk.numel()}"
        print(line, flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "_build":
        child_build()
    elif cmd == "_run":
        child_run(sys.argv[2])
    elif cmd == "build":
        procs = [subprocess.Popen([sys.executable, __file__, "_build"], env=env_for(d)) for _, d in PERTURB]
        sys.exit(max(p.wait() for p in procs))
    else:
        for name, d in PERTURB:
            r = subprocess.run([sys.executable, __file__, "_run", name], env=env_for(d), capture_output=True, text=True)
            print(r.stdout, end="", flush=True)
            if r.returncode:
                print(f"{name}: exit {r.returncode}: {r.stderr[-600:]}", flush=True)
