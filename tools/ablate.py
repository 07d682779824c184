"""Timing-only ablations of the lane-per-quartet J/K kernels (-DABL=<bits> builds of the development kernel sources,
joltqc_amd/csrc/kernels_dev): which part of a class kernel's time is the Rys table gather, the LDS atomics, the
density reads, the integral evaluation, and the staging / screening / flush around them.  Results are WRONG by
construction; only the per-class kernel times are read.

  python tools/ablate.py build            (CPU: compile every (class, ABL) code object into kcache_dev)
  python tools/ablate.py run [workload]   (GPU box) -> gpurun_out/ablate_<workload>.json
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLASSES = os.environ.get("JQC_ABL_CLASSES", "1010,2110,2111,3110,3120,3210").split(",")
ABLS = [int(x) for x in os.environ.get("JQC_ABL_SET", "0,1,2,4,8,16,7").split(",")]
DEV = os.path.join(ROOT, "joltqc_amd", "csrc", "kernels_dev")


def env_for(abl):
    e = dict(os.environ)
    e["JQC_KERNEL_SRC"] = DEV
    e["JQC_EXTRA_DEFS"] = f"-DABL={abl}"
    e["JQC_ONLY_CLASS"] = ",".join(CLASSES)
    e["JQC_STREAMS"] = "1"
    e["JQC_TRUST_KERNELS"] = "1"          # wrong by construction: no first-use cross-check
    return e


def _build_child():
    from joltqc_amd.backend import jk as router
    for key in CLASSES:
        ang = tuple(int(c) for c in key)
        router.gen_jk_kernel(ang, True, True, False, False, router.select_algo(ang), True)
        print("built", key, os.environ["JQC_EXTRA_DEFS"], flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "_child":
        _build_child()
    elif cmd == "build":
        procs = [subprocess.Popen([sys.executable, __file__, "_child"], env=env_for(a)) for a in ABLS]
        sys.exit(max(p.wait() for p in procs))
    else:
        wl = sys.argv[2] if len(sys.argv) > 2 else "0112-elongated-nitrogenous"
        out = {}
        for a in ABLS:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "class_profile.py"), wl], env=env_for(a),
                               capture_output=True, text=True)
            rows = {}
            for line in r.stdout.splitlines():
                if line.startswith("  ("):
                    ang = "".join(ch for ch in line.split(")")[0] if ch.isdigit())
                    rows[ang] = float(line.split(")")[1].split("ms")[0])
            out[str(a)] = rows
            print(f"ABL={a}: " + "  ".join(f"{k} {v:.2f}" for k, v in sorted(rows.items())), flush=True)
            if not rows:
                print(r.stdout[-2000:], r.stderr[-2000:])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"ablate_{wl}.json"), "w"), indent=1)
