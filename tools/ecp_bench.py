"""Timing of the device ECP integrals on a cluster of n^3 Na atoms (4.5 Bohr lattice) with the reference's test basis and type-2
potential (jqc/pyscf/tests/test_ecp_small.py:28-80).  usage: python tools/ecp_bench.py [n=2]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from joltqc_amd.backend import ecp as becp
from joltqc_amd.gto import mole
from test_ecp_oracle import BAS, ECP_TYPE2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
atoms = "; ".join(f"Na {4.5 * a} {4.5 * b} {4.5 * c}" for a in range(n) for b in range(n) for c in range(n))
mol = mole.Mole(atom=atoms, basis={"Na": BAS}, ecp={"Na": ECP_TYPE2}, unit="B")
for screen in (False, True):
    becp.get_ecp(mol, screen=screen); torch.cuda.synchronize()
    t = time.perf_counter(); h = becp.get_ecp(mol, screen=screen); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{n ** 3} atoms, nao {mol.nao}, screen={screen}: {becp.get_ecp.last_ntasks} tasks in {dt * 1e3:.1f} ms = {dt / becp.get_ecp.last_ntasks * 1e6:.1f} us/task, |h|max {float(h.abs().max()):.6f}")
