"""Large-molecule J/K run (stand-in geometries from joltqc_amd/data/molecules): timing + quartet counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.gto import mole
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
from joltqc_amd.roofline import quartet_flops
name, basis = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mol = mole.Mole(atom=mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
t0 = time.time(); lay = BasisLayout.from_mol(mol, alignment=tile_width)
print(f"{name}/{basis}: natm={mol.natm} nao={mol.nao} nbas(padded)={lay.nbasis} nao_int={lay.nao} layout {time.time()-t0:.2f}s", flush=True)
np.random.seed(9)
nocc = mol.nelectron // 2
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
for it in range(3):
    t = time.time(); vj, vk = g(mol, dm, hermi=1); torch.cuda.synchronize(); dt = time.time() - t
    print(f"  call {it}: {dt:.3f}s host {g.stats['host_seconds']:.3f}s launches {g.stats['launches']}", flush=True)
n64, n32, per = g.quartet_counts()
fl = sum(a * quartet_flops(ang, npr) for (ang, npr), (a, b) in per.items())
print(f"  quartets {n64:.4e}  {n64/dt:.3e} q/s  model {fl/1e12:.3f} TFLOP -> {fl/dt/1e12:.2f} TFLOP/s  |vj|max {float(vj.abs().max()):.3e} finite {bool(torch.isfinite(vj).all() and torch.isfinite(vk).all())}")
# ---- size-independent parity properties at full size (no CPU oracle finishes here in seconds):
#  (1) launch geometry: ket chunks of 1 vs the default, no workgroup splits -> same J/K
#  (2) independent algorithm: the queue-driven one-quartet-per-lane kernels (jk_1q1t.hip, direct global atomics)
#  (3) symmetry of J and K for a symmetric density;  (4) linearity in the density
if len(sys.argv) > 3 and sys.argv[3] == "check":
    ref_j, ref_k = vj.clone(), vk.clone()
    sc = float(max(ref_j.abs().max(), ref_k.abs().max()))
    jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = 1, 1
    g2 = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    j2, k2 = g2(mol, dm, hermi=1)
    print(f"  (1) kchunk=1 vs default:   dJ {float((j2-ref_j).abs().max())/sc:.2e}  dK {float((k2-ref_k).abs().max())/sc:.2e}  (relative to max element)")
    jkmod.KCHUNK_MAX, jkmod.NSPLIT_MAX = 16, 8
    os.environ["JQC_JK_ALGO"] = "1q1t"
    from joltqc_amd.backend import jk as router
    router.gen_jk_kernel.cache_clear()
    t = time.time()
    g3 = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    j3, k3 = g3(mol, dm, hermi=1); torch.cuda.synchronize()
    print(f"  (2) queue 1q1t kernels ({time.time()-t:.1f}s incl. JIT): dJ {float((j3-ref_j).abs().max())/sc:.2e}  dK {float((k3-ref_k).abs().max())/sc:.2e}")
    del os.environ["JQC_JK_ALGO"]; router.gen_jk_kernel.cache_clear()
    print(f"  (3) asymmetry: J {float((ref_j-ref_j.T).abs().max())/sc:.2e}  K {float((ref_k-ref_k.T).abs().max())/sc:.2e}")
    c2 = np.random.rand(mol.nao, nocc) - 0.5
    dm2 = torch.from_numpy(c2 @ c2.T / nocc).cuda()
    ja, ka = g(mol, dm2, hermi=1)
    jb, kb = g(mol, dm + 0.5 * dm2, hermi=1)
    print(f"  (4) linearity: J {float((jb-ref_j-0.5*ja).abs().max())/sc:.2e}  K {float((kb-ref_k-0.5*ka).abs().max())/sc:.2e}")
    # (5) long-range (erf-attenuated) K with the tiled kernels vs the queue kernels; (6) mixed precision vs pure fp64
    kj, kk = g(mol, dm, hermi=1, omega=0.3)
    os.environ["JQC_JK_ALGO"] = "1q1t"; router.gen_jk_kernel.cache_clear()
    g4 = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    qj, qk = g4(mol, dm, hermi=1, omega=0.3)
    del os.environ["JQC_JK_ALGO"]; router.gen_jk_kernel.cache_clear()
    print(f"  (5) omega=0.3 tiled vs queue: dJ {float((kj-qj).abs().max())/sc:.2e}  dK {float((kk-qk).abs().max())/sc:.2e}  |K_lr|max {float(kk.abs().max()):.3e}")
    gm = jkmod.generate_jk_kernel(lay, cutoff_fp64=1e-7, cutoff_fp32=1e-13)
    mj, mk = gm(mol, dm, hermi=1)
    n64m, n32m, _ = gm.quartet_counts()
    print(f"  (6) mixed 1e-13/1e-7 vs fp64: dJ {float((mj-ref_j).abs().max()):.2e}  dK {float((mk-ref_k).abs().max()):.2e} (absolute; reference bar 1e-7)  fp64 quartets {n64m:.3e} fp32 quartets {n32m:.3e}")
