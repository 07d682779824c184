"""Compile the tile kernels (s..f, J+K, fp64) with extra defines into joltqc_amd/csrc/kcache_<name>
(travels to the GPU box).  usage: python tools/build_variant.py <name> "<defs>" [algo]"""
import os, sys
name, defs = sys.argv[1], sys.argv[2]
algo = int(sys.argv[3]) if len(sys.argv) > 3 else 1
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["JQC_EXTRA_DEFS"] = defs
os.environ["JQC_KERNEL_CACHE"] = os.path.join(root, "joltqc_amd", "csrc", "kcache_" + name)
sys.path.insert(0, root)
from multiprocessing import get_context
import __graft_entry__ as g
if __name__ == "__main__":
    from joltqc_amd.backend import lib as L
    L.lib()
    jobs = [(ang, 1, 1, 0, 0, algo) for ang in g._classes(3)]
    with get_context("spawn").Pool(8) as pool:
        errs = [e for e in pool.imap_unordered(g._compile_one, jobs) if e]
    print(name, "errors:", errs)
