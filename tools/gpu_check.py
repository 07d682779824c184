"""Quick GPU sanity run: J/K parity vs the CPU oracle on small molecules + a first timing.
Logs progressively to gpurun_out/check.log."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joltqc_amd.gto import mole
from joltqc_amd.pyscf.basis import BasisLayout
from joltqc_amd.pyscf import jk as jkmod
from oracle import dense

os.makedirs("gpurun_out", exist_ok=True)
LOG = open("gpurun_out/check.log", "a")
def P(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); LOG.write(s + "\n"); LOG.flush()

H2O = "O 0 0 0.1174; H -0.757 0 -0.4696; H 0.757 0 -0.4696"

def benzene():
    rc, rh = 1.39, 1.39 + 1.09
    a = []
    for k in range(6):
        t = np.pi / 3 * k
        a.append(("C", (rc * np.cos(t), rc * np.sin(t), 0.0)))
        a.append(("H", (rh * np.cos(t), rh * np.sin(t), 0.0)))
    return a

def check(name, atom, basis, cart=False, omega=None, hermi=1, oracle=True, unit="angstrom", reps=3):
    mol = mole.Mole(atom=atom, basis=basis, cart=cart, unit=unit)
    from joltqc_amd.constants import tile_width
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    nao = mol.nao
    dm = np.random.rand(nao, nao)
    dm = dm @ dm.T if hermi == 1 else dm
    get_jk = jkmod.generate_jk_kernel(lay, cutoff_fp64=1e-13, cutoff_fp32=1e-13)
    P(f"{name}: nao={nao} nbas={lay.nbasis} groups={lay.group_key.tolist()}")
    t0 = time.time()
    vj, vk = get_jk(mol, dm, hermi=hermi, omega=omega)
    torch.cuda.synchronize()
    t1 = time.time()
    P(f"   first call {t1-t0:.2f}s")
    ts = []
    for _ in range(reps):
        t = time.time(); vj, vk = get_jk(mol, dm, hermi=hermi, omega=omega); torch.cuda.synchronize(); ts.append(time.time() - t)
    n64, n32, per = get_jk.quartet_counts()
    P(f"   steady={min(ts)*1e3:.2f}ms quartets={n64} ({n64/min(ts):.3e}/s) launches={get_jk.stats['launches']} host={get_jk.stats['host_seconds']*1e3:.1f}ms")
    if oracle:
        t = time.time()
        rj, rk = dense.get_jk(lay, dm, hermi=hermi, omega=omega)
        P(f"   oracle {time.time()-t:.1f}s  max|dJ|={np.abs(vj.cpu().numpy()-rj).max():.3e} max|dK|={np.abs(vk.cpu().numpy()-rk).max():.3e}  |J|max={np.abs(rj).max():.3e}")
    return get_jk, mol, dm

if __name__ == "__main__":
    P(torch.cuda.get_device_name(0))
    which = sys.argv[1:] or ["h2", "h2o", "lr", "h0", "bsvp", "btz"]
    if "h2" in which: check("H2/tzvpp cart", "H -0.757 4. -0.4696; H 0.757 4. -0.4696", "def2-tzvpp", True, unit="B")
    if "h2o" in which: check("H2O/tzvpp sph", H2O, "def2-tzvpp")
    if "lr" in which: check("H2O/svp sph omega=0.3", H2O, "def2-svp", omega=0.3)
    if "h0" in which: check("H2O/svp sph hermi=0", H2O, "def2-svp", hermi=0)
    if "bsvp" in which: check("benzene/svp", benzene(), "def2-svp", oracle=True)
    if "btz" in which:
        g, mol, dm = check("benzene/tzvpp", benzene(), "def2-tzvpp", oracle=False)
        n64, n32, per = g.quartet_counts()
        for k in sorted(per): P("     ", k, per[k][0])
