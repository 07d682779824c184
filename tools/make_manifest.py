"""Write joltqc_amd/data/verified_kernels.json: the class-kernel builds covered by the gates of the scheme table
(= the AOT set of __graft_entry__.kernel_jobs), for the current kernel sources and compiler (source tag).

Run it ONLY after `pytest tests -m gpu` is green on an MI355X for exactly these sources: the gates are
tests/test_jk_gpu.py::test_every_angular_class_against_the_oracle (all 140 classes x modes x both scheme tables),
::test_every_kernel_variant_of_the_scheme_table, and tests/test_jk_fullsize_gpu.py (forced ket chunks on the s..g
benzene, 112 atoms tiled vs queue kernels).  A build that is not listed is cross-checked on first use
(joltqc_amd/pyscf/jk.py: first_use_check).
usage: python tools/make_manifest.py "<evidence: which GPU run was green>"
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
from joltqc_amd.backend import jk as router, lib as L

keys = set()
for ang, dj, dk, lr, fp32, algo in G.kernel_jobs():
    if (algo & 0xf) == L.ALGO_1Q1T:
        continue
    router.gen_jk_kernel(ang, bool(dj), bool(dk), bool(lr), bool(fp32), algo, True)      # cached code object: no compile
    built = router.resolved_algo(ang, dj, dk, lr, fp32, algo)
    keys.add(router.kernel_key(ang, dj, dk, lr, fp32, built))
out = {"_comment": "class-kernel builds that passed the GPU gates (tools/make_manifest.py); unlisted builds are cross-checked "
                   "against the one-quartet-per-lane kernel on first use",
       "src_tag": L.lib().jqc_source_tag().decode(), "evidence": sys.argv[1] if len(sys.argv) > 1 else "",
       "keys": sorted(keys)}
path = os.path.join(ROOT, "joltqc_amd", "data", "verified_kernels.json")
json.dump(out, open(path, "w"), indent=0)
print(f"{len(keys)} builds, source tag {out['src_tag']} -> {path}")
