"""Diagnostics for DESIGN.md 3.1 (wrong-result builds): do the banned builds of the row-lane kernels -- one wave per SIMD (more than
256 registers per lane), the double-buffered TRR schedule -- still give wrong J/K with the round-3 sources?
  build (CPU): compile the variants of every class s..g into the kernel cache of each configuration
  run   (GPU): tools/verify_scheme.py on benzene with the artificial s..g basis and forced ket chunks, per configuration
usage: python tools/diag_banned_builds.py build|run"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (name, variant code forced on every class, extra definitions)
CONFIGS = [("minw1_rowlane", 0x111, ""), ("minw1_cjr", 0x911, ""), ("minw1_wsync", 0x511, ""),
           ("dbuf_rowlane", 0x121, "-DTRR_DOUBLE_BUFFER=1"), ("dbuf_cjr", 0x921, "-DTRR_DOUBLE_BUFFER=1"),
           ("minw1_dbuf", 0x111, "-DTRR_DOUBLE_BUFFER=1"), ("control_minw2", 0x121, "")]
CACHE = os.path.join(ROOT, "joltqc_amd", "csrc", "kcache_diag")


def env_for(v, defs):
    e = dict(os.environ)
    e.update(JQC_KERNEL_CACHE=CACHE, JQC_EXTRA_DEFS=defs, JQC_TRUST_KERNELS="1", JQC_VERIFY_ALGO=hex(v), JQC_TARGET_WGS="32",
             JQC_KCHUNK_MAX="8", JQC_STREAMS="1")
    return e


def child_build(v):
    from joltqc_amd.backend import jk as router
    classes = [(a, b, c, d) for a in range(5) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]
    bad = 0
    for ang in classes:
        for algo in (router.forced_variant(ang, v), 0x221):
            try:
                router.gen_jk_kernel(ang, True, True, False, False, algo, True)
            except RuntimeError:
                bad += 1
    print("built variant", hex(v), "defs", repr(os.environ.get("JQC_EXTRA_DEFS")), "failures", bad, flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "_build":
        child_build(int(sys.argv[2], 0))
    elif cmd == "build":
        procs = [subprocess.Popen([sys.executable, __file__, "_build", hex(v)], env=env_for(v, d)) for _, v, d in CONFIGS]
        sys.exit(max(p.wait() for p in procs))
    else:
        os.makedirs(os.path.join(ROOT, "gpurun_out", "r03_diag_banned"), exist_ok=True)
        for name, v, d in CONFIGS:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_scheme.py"), "benzene-spdfg", "1e-10"],
                               env=env_for(v, d), capture_output=True, text=True)
            lines = [l for l in r.stdout.splitlines() if "MISMATCH" in l]
            open(os.path.join(ROOT, "gpurun_out", "r03_diag_banned", name + ".txt"), "w").write(r.stdout + r.stderr[-3000:])
            print(f"{name} (variant {v:#x}, defs {d!r}): {len(lines)} mismatching lines; exit {r.returncode}", flush=True)
            for l in lines[:12]:
                print("   ", l, flush=True)
