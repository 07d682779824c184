#!/bin/bash
# round 3, A/B 7: ket chunk <-> XCD affinity (chunk counts and first blocks of the task rows rounded to multiples of 8), all 65 classes,
# development kernels (skip of ket pairs without canonical quartets included)
export HSA_ENABLE_IPC_MODE_LEGACY=0
export JQC_KERNEL_SRC=$PWD/joltqc_amd/csrc/kernels_dev JQC_TRUST_KERNELS=1 JQC_STREAMS=1
O=gpurun_out/r03_ab7; mkdir -p $O
for a in 1 8 1 8; do
  JQC_CHUNK_ALIGN=$a timeout 600 python tools/class_profile.py 0112-elongated-nitrogenous > $O/align${a}_$RANDOM.txt 2>&1
done
grep -H "total serial" $O/*.txt
# end-to-end step (4 streams) with and without
for a in 1 8; do
  JQC_CHUNK_ALIGN=$a JQC_STREAMS= timeout 600 python - <<'PY' >> $O/step.txt 2>&1
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.pop("JQC_STREAMS", None)
import numpy as np, torch
from bench import load_workload
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
from joltqc_amd.pyscf.basis import BasisLayout
mol, name = load_workload("0112-elongated-nitrogenous")
lay = BasisLayout.from_mol(mol, alignment=tile_width)
np.random.seed(9)
dm = np.random.rand(mol.nao, mol.nao); dm = torch.from_numpy(dm @ dm.T).cuda()
g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
for _ in range(2): g(mol, dm, hermi=1)
torch.cuda.synchronize(); t = time.time()
for _ in range(4): g(mol, dm, hermi=1)
torch.cuda.synchronize()
print("CHUNK_ALIGN", os.environ.get("JQC_CHUNK_ALIGN"), "step ms", (time.time() - t) / 4 * 1e3, "quartets", g.quartet_counts()[0])
PY
done
cat $O/step.txt
