"""Write joltqc_amd/data/verified_kernels.json: the class-kernel builds covered by the gates of the scheme table
(= the AOT set of __graft_entry__.kernel_jobs), for the current kernel sources and compiler (source tag).

Run it ONLY after `pytest tests -m gpu` is green on an MI355X for exactly these sources: the gates are
tests/test_jk_gpu.py::test_every_angular_class_against_the_oracle (all 140 classes x modes x both scheme tables),
::test_every_kernel_variant_of_the_scheme_table, and tests/test_jk_fullsize_gpu.py (forced ket chunks on the s..g
benzene, 112 atoms tiled vs queue kernels).  A build that is not listed is cross-checked on first use
(joltqc_amd/pyscf/jk.py: first_use_check).
QUARANTINE RULE (round 4): a build with more than 256 registers per lane AND SGPRs spilled to VGPR lanes -- the one family of
wrong-result builds ever found -- is listed only with a passing record of tools/risky_builds_gate.py for THIS source
tag (joltqc_amd/data/risky_builds_gate.json: the forced-ket-chunk gate twice, run-to-run agreement <= 1e-12); otherwise it goes to
"quarantined" and stays subject to the first-use cross-check.  The register / spill figures of every risky build are recorded.
usage: python tools/make_manifest.py "<evidence: which GPU run was green>"
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as G
from joltqc_amd.backend import jk as router, lib as L

keys = set()
for ang, dj, dk, lr, fp32, algo in G.kernel_jobs():
    if (algo & 0xf) == L.ALGO_1Q1T:
        continue
    router.gen_jk_kernel(ang, bool(dj), bool(dk), bool(lr), bool(fp32), algo, True)      # cached code object: no compile
    built = router.resolved_algo(ang, dj, dk, lr, fp32, algo)
    keys.add(router.kernel_key(ang, dj, dk, lr, fp32, built))
import risky_builds_gate as RG
risky, _all = RG.risky_builds()
gate_path = os.path.join(ROOT, "joltqc_amd", "data", "risky_builds_gate.json")
gate = json.load(open(gate_path)) if os.path.exists(gate_path) else {}
tag_now = L.lib().jqc_source_tag().decode()
passed = {k for k, v in gate.get("results", {}).items() if v.get("ok") and v["run_to_run"] <= 1e-12} if gate.get("src_tag") == tag_now else set()
quarantined = sorted(k for k in keys if k in risky and k not in passed)
keys -= set(quarantined)
print(f"{len(risky)} risky builds (> 256 registers and SGPR spills): {len(risky) - len(quarantined)} passed the twice-run gate, "
      f"{len(quarantined)} quarantined")
out = {"_comment": "class-kernel builds that passed the GPU gates (tools/make_manifest.py); unlisted builds are cross-checked "
                   "against the one-quartet-per-lane kernel on first use",
       "src_tag": L.lib().jqc_source_tag().decode(), "evidence": sys.argv[1] if len(sys.argv) > 1 else "",
       "risky": {k: [risky[k][0], risky[k][2]] for k in sorted(risky)},        # {key: [registers per lane, SGPRs spilled to lanes]}
       "quarantined": quarantined,
       "keys": sorted(keys)}
path = os.path.join(ROOT, "joltqc_amd", "data", "verified_kernels.json")
json.dump(out, open(path, "w"), indent=0)
print(f"{len(keys)} builds, source tag {out['src_tag']} -> {path}")
