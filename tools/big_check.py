"""Large-molecule J/K run (stand-in geometries from joltqc_amd/data/molecules): timing + quartet counts, and with
``check`` the size-independent parity properties at full size (no CPU oracle finishes there in seconds):
  (1) launch geometry: ket chunks of 1 vs the default, no workgroup splits -> same J/K
  (2) independent algorithm: the queue-driven one-quartet-per-lane kernels (jqc_screen_jk_tasks + jk_1q1t.hip, direct
      global atomics)
  (3) symmetry of J and K for a symmetric density;  (4) linearity in the density
  (5) long-range (erf-attenuated) J/K, tiled vs queue kernels;  (6) mixed precision vs pure fp64
usage: python tools/big_check.py <xyz name> <basis> [check]; tests/test_jk_fullsize_gpu.py calls ``check``."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def setup(name, basis):
    import numpy as np, torch
    from joltqc_amd.gto import mole
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf.basis import BasisLayout
    mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    nocc = mol.nelectron // 2
    c = np.random.rand(mol.nao, nocc) - 0.5
    dm = torch.from_numpy(c @ c.T / nocc).cuda()
    return mol, lay, dm


def check(mol, lay, dm, g=None, log=print, lr=True):
    """The six properties; returns {name: error} (relative to the largest J/K element except ``mixed``: absolute)."""
    import numpy as np, torch
    from joltqc_amd.backend import jk as router
    from joltqc_amd.pyscf import jk as jkmod
    res = {}
    g = g or jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    ref_j, ref_k = (x.clone() for x in g(mol, dm, hermi=1))
    sc = float(max(ref_j.abs().max(), ref_k.abs().max()))
    rel = lambda a, b: float((a - b).abs().max()) / sc
    kc, ns = jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX
    saved = os.environ.get("JQC_JK_ALGO")
    try:
        jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = 1, 1
        j2, k2 = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)(mol, dm, hermi=1)
        res["chunk_J"], res["chunk_K"] = rel(j2, ref_j), rel(k2, ref_k)
        log(f"  (1) kchunk=1 vs default:   dJ {res['chunk_J']:.2e}  dK {res['chunk_K']:.2e}  (relative to max element)")
        jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = kc, ns
        os.environ["JQC_JK_ALGO"] = "1q1t"
        router.gen_jk_kernel.cache_clear()
        t = time.time()
        g3 = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
        j3, k3 = g3(mol, dm, hermi=1); torch.cuda.synchronize()
        res["queue_J"], res["queue_K"] = rel(j3, ref_j), rel(k3, ref_k)
        res["queue_n"], res["tile_n"] = g3.quartet_counts()[0], g.quartet_counts()[0]
        log(f"  (2) queue 1q1t kernels ({time.time()-t:.1f}s): dJ {res['queue_J']:.2e}  dK {res['queue_K']:.2e}  quartets {res['queue_n']} vs tiled {res['tile_n']}")
        if lr:
            qj, qk = g3(mol, dm, hermi=1, omega=0.3)
        del os.environ["JQC_JK_ALGO"]; router.gen_jk_kernel.cache_clear()
        res["asym_J"], res["asym_K"] = rel(ref_j, ref_j.T), rel(ref_k, ref_k.T)
        log(f"  (3) asymmetry: J {res['asym_J']:.2e}  K {res['asym_K']:.2e}")
        nocc = max(mol.nelectron // 2, 1)
        c2 = np.random.rand(mol.nao, nocc) - 0.5
        dm2 = torch.from_numpy(c2 @ c2.T / nocc).cuda()
        ja, ka = g(mol, dm2, hermi=1)
        jb, kb = g(mol, dm + 0.5 * dm2, hermi=1)
        res["lin_J"], res["lin_K"] = rel(jb, ref_j + 0.5 * ja), rel(kb, ref_k + 0.5 * ka)
        log(f"  (4) linearity: J {res['lin_J']:.2e}  K {res['lin_K']:.2e}")
        if lr:
            kj, kk = g(mol, dm, hermi=1, omega=0.3)
            res["lr_J"], res["lr_K"] = rel(kj, qj), rel(kk, qk)
            res["lr_Kmax"] = float(kk.abs().max())
            log(f"  (5) omega=0.3 tiled vs queue: dJ {res['lr_J']:.2e}  dK {res['lr_K']:.2e}  |K_lr|max {res['lr_Kmax']:.3e}")
        gm = jkmod.generate_jk_kernel(lay, cutoff_fp64=1e-7, cutoff_fp32=1e-13)
        mj, mk = gm(mol, dm, hermi=1)
        n64m, n32m, _ = gm.quartet_counts()
        res["mixed_J"], res["mixed_K"] = float((mj - ref_j).abs().max()), float((mk - ref_k).abs().max())
        log(f"  (6) mixed 1e-13/1e-7 vs fp64: dJ {res['mixed_J']:.2e}  dK {res['mixed_K']:.2e} (absolute; reference bar 1e-7)  fp64 quartets {n64m:.3e} fp32 quartets {n32m:.3e}")
    finally:
        jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = kc, ns
        if saved is None:
            os.environ.pop("JQC_JK_ALGO", None)
        else:
            os.environ["JQC_JK_ALGO"] = saved
        router.gen_jk_kernel.cache_clear()
    res["_ref"] = (ref_j, ref_k, g)
    return res


if __name__ == "__main__":
    import torch
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.roofline import quartet_flops
    name, basis = sys.argv[1], sys.argv[2]
    t0 = time.time(); mol, lay, dm = setup(name, basis)
    print(f"{name}/{basis}: natm={mol.natm} nao={mol.nao} nbas(padded)={lay.nbasis} nao_int={lay.nao} layout {time.time()-t0:.2f}s", flush=True)
    g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    for it in range(3):
        t = time.time(); vj, vk = g(mol, dm, hermi=1); torch.cuda.synchronize(); dt = time.time() - t
        print(f"  call {it}: {dt:.3f}s host {g.stats['host_seconds']:.3f}s launches {g.stats['launches']}", flush=True)
    n64, n32, per = g.quartet_counts()
    fl = sum(a * quartet_flops(ang, npr) for (ang, npr), (a, b) in per.items())
    print(f"  quartets {n64:.4e}  {n64/dt:.3e} q/s  model {fl/1e12:.3f} TFLOP -> {fl/dt/1e12:.2f} TFLOP/s  |vj|max {float(vj.abs().max()):.3e} finite {bool(torch.isfinite(vj).all() and torch.isfinite(vk).all())}")
    if len(sys.argv) > 3 and sys.argv[3] == "check":
        check(mol, lay, dm, g)
