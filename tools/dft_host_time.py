"""Where a rho / vxc call of the bench's grid leg spends its wall time: wall per call, host-only time (calls issued without a
sync), cProfile of the host side, distribution of the significant-AO count per block.  Run under
`rocprofv3 --kernel-trace --stats` for the kernel side.    usage: python tools/dft_host_time.py [workload] [xc]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import load_workload
from joltqc_amd.gto.grids import Grids
from joltqc_amd.pyscf import rks
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload(sys.argv[1] if len(sys.argv) > 1 else "0112-elongated-nitrogenous")
xc = sys.argv[2] if len(sys.argv) > 2 else "GGA"
ndim = {"LDA": 1, "GGA": 4, "MGGA": 5}[xc]
lay = BasisLayout.from_mol(mol, alignment=1)
gg = Grids(mol, 30, 8).build()
order = rks.arg_group_grids(gg.coords)
n = gg.coords.shape[0] // 256 * 256
class G: pass
g = G(); g.coords = gg.coords[order][:n]; g.weights = gg.weights[order][:n]
_, rho_k, vxc_k = rks.generate_rks_kernel(lay)
np.random.seed(9)
nocc = max(mol.nelectron // 2, 1)
c = np.random.rand(mol.nao, nocc) - 0.5
dm = torch.from_numpy(c @ c.T / nocc).cuda()
wv = torch.rand((ndim, n), dtype=torch.float64, device="cuda")
for fn, arg, label in ((rho_k, dm, "rho"), (vxc_k, wv, "vxc")):
    fn(mol, g, xc, arg); torch.cuda.synchronize()
    t = time.perf_counter(); fn(mol, g, xc, arg); th = time.perf_counter() - t; torch.cuda.synchronize(); tw = time.perf_counter() - t
    N = 5
    t = time.perf_counter()
    for _ in range(N): fn(mol, g, xc, arg)
    th5 = (time.perf_counter() - t) / N; torch.cuda.synchronize(); tw5 = (time.perf_counter() - t) / N
    m = rho_k.stats["nrow_h"].astype(float)
    print(f"{label} {xc}: one call host {th*1e3:.2f} ms wall {tw*1e3:.2f} ms; back-to-back host {th5*1e3:.2f} wall {tw5*1e3:.2f} ms/call; "
          f"blocks {m.size} m mean {m.mean():.0f} median {np.median(m):.0f} max {m.max():.0f} p90 {np.percentile(m, 90):.0f} "
          f"sum m^2 {float((m*m).sum()):.3e} (mean m)^2 n {m.mean()**2*m.size:.3e}", flush=True)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): fn(mol, g, xc, arg)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
