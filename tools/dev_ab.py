"""A/B harness for kernel development: builds of the development kernel sources (joltqc_amd/csrc/kernels_dev) under
several -D option sets are (1) checked against the CPU oracle class by class on a three-atom s..f system and (2) timed
per class on a large molecule.  The shipped sources and their verified code objects are not touched.

  python tools/dev_ab.py build  <classes> <name=defs> [<name=defs> ...]     (CPU: compile into kcache_dev)
  python tools/dev_ab.py run    <classes> <name=defs> [...]                 (GPU box) -> gpurun_out/dev_ab_<tag>.json
classes: comma list such as 2110,3120 or "tile1q" (every class the scheme table routes to the lane-per-quartet mode) or
"all" (s..f).  defs example:  base=  qil=-DQIL=1  all="-DQIL=1 -DCORD=1".  JQC_AB_ALGO=<code> forces one variant code for every
configuration; a configuration written  name=@<code>:<defs>  forces its own (e.g. quad=@0x1001022:).
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.path.join(ROOT, "joltqc_amd", "csrc", "kernels_dev")


def class_list(spec):
    from joltqc_amd.backend import jk as router
    allc = [(a, b, c, d) for a in range(4) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]
    if spec == "all":
        return allc
    if spec == "tile1q":
        return [a for a in allc if (router.select_algo(a) & 0xf) == 2]
    if spec == "rowlane":
        return [a for a in allc if (router.select_algo(a) & 0xf) == 1]
    return [tuple(int(ch) for ch in k) for k in spec.split(",")]


def env_for(defs, classes):
    e = dict(os.environ)
    if defs.startswith("@"):
        code, defs = defs[1:].split(":", 1)
        e["JQC_JK_ALGO"] = "v%d" % int(code, 0)
    e.update(JQC_KERNEL_SRC=DEV, JQC_EXTRA_DEFS=defs, JQC_TRUST_KERNELS="1", JQC_STREAMS="1",
             JQC_ONLY_CLASS=",".join("%d%d%d%d" % a for a in classes))
    if os.environ.get("JQC_AB_ALGO"):
        e["JQC_JK_ALGO"] = "v%d" % int(os.environ["JQC_AB_ALGO"], 0)
    return e


def ok_file():
    import hashlib
    h = hashlib.md5((os.environ.get("JQC_EXTRA_DEFS", "") + "|" + os.environ.get("JQC_JK_ALGO", "")).encode()).hexdigest()[:12]
    return os.path.join(ROOT, "joltqc_amd", "csrc", "kcache_dev", f"ok_{h}.json")


def child_build(classes):
    """Compile; the classes whose build exists (no LDS overflow, no static assertion) are recorded next to the code objects, and a
    timing run of this configuration is restricted to them."""
    from joltqc_amd.backend import jk as router
    bad, ok = [], []
    for ang in classes:
        good = True
        for small in (False, True):
            try:
                want = router.select_algo(ang, small=small)
                router.gen_jk_kernel(ang, True, True, False, False, want, True)
                if os.environ.get("JQC_JK_ALGO") and router.resolved_algo(ang, True, True, False, False, want) != want:
                    good = False                      # (the router fell back to a smaller form: not the configuration asked for)
            except RuntimeError as e:
                good = False
                bad.append(("%d%d%d%d" % ang, str(e).strip().splitlines()[-1][-60:] if "static assertion" not in str(e) else "static_assert"))
        if good:
            ok.append("%d%d%d%d" % ang)
    json.dump(ok, open(ok_file(), "w"))
    print("built", len(ok), "of", len(classes), "classes with", repr(os.environ.get("JQC_EXTRA_DEFS")), os.environ.get("JQC_JK_ALGO", ""),
          "failed:", " ".join(b[0] for b in bad), flush=True)


def child_check(classes):
    """Every class vs the oracle on C2H with an artificial s..f basis (the all-class gate of tests/test_jk_gpu.py)."""
    import numpy as np
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
              [2, [0.8, 1.0]], [3, [0.9, 1.0]]]
    mol = mole.Mole(atom="C 0 0 0; C 0 0.3 2.4; H 1.5 0.2 0.9", basis={"C": shells, "H": shells}, unit="B")
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao); dm = dm @ dm.T
    allq = dense.canonical_quartets(lay)
    qa = np.asarray(lay.angs)[allq.astype(int)]
    worst = 0.0
    bad = []
    for target in (4096, 1):                     # small-launch table, then main table with long ket chunks
        jkmod.TARGET_WGS = target
        for ang in classes:
            sel = (qa == np.array(ang)).all(1)
            if not sel.any():
                continue
            rj, rk = dense.get_jk(lay, dm, hermi=1, quartets=allq[sel])
            os.environ["JQC_ONLY_CLASS"] = "%d%d%d%d" % ang
            g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
            vj, vk = g(mol, dm, hermi=1)
            sc = max(np.abs(rj).max(), np.abs(rk).max())
            err = max(np.abs(vj.cpu().numpy() - rj).max(), np.abs(vk.cpu().numpy() - rk).max()) / sc
            n = g.quartet_counts()[0]
            worst = max(worst, err)
            if not err < 1e-11 or n != int(sel.sum()):
                bad.append(("%d%d%d%d" % ang, target, err, n, int(sel.sum())))
    print(json.dumps({"worst": worst, "bad": bad}), flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd in ("_build", "_check"):
        cl = [tuple(int(ch) for ch in k) for k in sys.argv[2].split(",")]
        (child_build if cmd == "_build" else child_check)(cl)
        sys.exit(0)
    spec = sys.argv[2]
    cfgs = [(x.split("=", 1)[0], x.split("=", 1)[1]) for x in sys.argv[3:]]
    classes = class_list(spec)
    spec = ",".join("%d%d%d%d" % a for a in classes)
    if cmd == "build":
        rc = 0
        for lo in range(0, len(cfgs), 4):          # four configurations at a time (each child compiles serially)
            procs = [subprocess.Popen([sys.executable, __file__, "_build", spec], env=env_for(d, classes)) for _, d in cfgs[lo:lo + 4]]
            rc = max([rc] + [p.wait() for p in procs])
        sys.exit(rc)
    wl = os.environ.get("JQC_AB_WORKLOAD", "0112-elongated-nitrogenous")
    out = {}
    for name, d in cfgs:
        env = env_for(d, classes)
        okf = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import dev_ab; print(dev_ab.ok_file())" % os.path.dirname(os.path.abspath(__file__))],
                             env=env, capture_output=True, text=True).stdout.strip()
        if okf and os.path.exists(okf):                  # classes that built for this configuration (recorded by `build`)
            good = set(json.load(open(okf)))
            env["JQC_ONLY_CLASS"] = ",".join(k for k in env["JQC_ONLY_CLASS"].split(",") if k in good)
            if not env["JQC_ONLY_CLASS"]:
                out[name] = {"defs": d, "check": "no class built", "ms": {}, "sum_ms": 0.0}
                continue
        if os.environ.get("JQC_AB_NOCHECK"):             # timing sweep: the winners are checked in a second run
            chk = "skipped"
        else:
            r = subprocess.run([sys.executable, __file__, "_check", env["JQC_ONLY_CLASS"]], env=env, capture_output=True, text=True)
            chk = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "class_profile.py"), wl], env=env, capture_output=True, text=True)
        rows = {}
        for line in r.stdout.splitlines():
            if line.startswith("  ("):
                ang = "".join(ch for ch in line.split(")")[0] if ch.isdigit())
                rows[ang] = float(line.split(")")[1].split("ms")[0])
        out[name] = {"defs": d, "check": chk, "ms": rows, "sum_ms": sum(rows.values())}
        print(f"{name:10s} sum {sum(rows.values()):9.2f} ms  check {chk[:160]}", flush=True)
        if not rows:
            print(r.stdout[-1500:], r.stderr[-1500:])
    base = out[cfgs[0][0]]["ms"]
    for k in sorted(base):
        print(k, "  ".join(f"{n} {out[n]['ms'].get(k, float('nan')):8.2f}" for n, _ in cfgs))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"dev_ab_{os.environ.get('JQC_AB_TAG', 'run')}.json"), "w"), indent=1)
