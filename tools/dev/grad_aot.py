"""AOT-compile the two-electron gradient kernels of the s..f classes in ONE form (JQC_GRAD_COOP=0|1 from the environment) into kcache,
so a per-class A/B on the GPU box (tools/grad_ab.py) does not spend its minutes in hiprtc.   usage: JQC_GRAD_COOP=0 python tools/dev/grad_aot.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiprocessing import get_context


def one(cls):
    from joltqc_amd.backend import lib as L
    t = time.time()
    try:
        L.check(L.lib().jqc_gen_jk_grad_kernel(*cls, 0, 1))
        return cls, time.time() - t, None
    except Exception as e:  # noqa: BLE001
        return cls, time.time() - t, str(e)


if __name__ == "__main__":
    if len(sys.argv) > 1 and len(sys.argv[1]) == 4:
        canon = [tuple(int(x) for x in c) for c in sys.argv[1:]]
    else:
        lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 3
        canon = [(a, b, c, d) for a in range(lmax + 1) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]
    canon.sort(key=lambda c: -sum(c))
    t0 = time.time()
    with get_context("spawn").Pool(8) as pool:
        for cls, dt, err in pool.imap_unordered(one, canon, chunksize=1):
            if err or dt > 60:
                print(cls, f"{dt:.0f}s", err or "", flush=True)
    print(f"done {len(canon)} classes in {time.time() - t0:.0f}s", flush=True)
