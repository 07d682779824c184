"""Ahead-of-time compile (CPU, 8 processes) of the lane-per-quartet classes' builds under the current kernel sources: the scheme
table's variant and its fused mixed-precision form (JQC_VARIANT_MIXED), J+K / J / K, with and without range separation.
usage: [JQC_KERNEL_SRC=...] python tools/build_mixed_dev.py [jk|all]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES_ALL = [(1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 1, 1)]


def job(args):
    a, modes = args
    from joltqc_amd.backend import jk as router
    out = []
    for small in (False, True):
        v = router.select_algo(a, small=small)
        if (v & 0xf) != 2:
            continue
        for code in (v, router.mixed_variant(a, v)):
            for dj, dk, lr in modes:
                try:
                    router.gen_jk_kernel(a, bool(dj), bool(dk), bool(lr), False, code, True)
                except Exception as e:  # noqa: BLE001
                    out.append((a, code, str(e)[-200:]))
    return out


if __name__ == "__main__":
    from multiprocessing import get_context
    modes = MODES_ALL if (len(sys.argv) > 1 and sys.argv[1] == "all") else [(1, 1, 0)]
    ALL = [(a, b, c, d) for a in range(4) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]
    t = time.time()
    with get_context("spawn").Pool(8) as p:
        bad = [x for r in p.imap_unordered(job, [(a, modes) for a in ALL], chunksize=1) for x in r]
    print("failures", bad)
    print("time", round(time.time() - t, 1))
