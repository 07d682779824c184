#!/bin/bash
# stamps of the owner-reduction builds (development sources) of the dominant j-in-registers classes
export HSA_ENABLE_IPC_MODE_LEGACY=0
export JQC_KERNEL_SRC=$PWD/joltqc_amd/csrc/kernels_dev
mkdir -p gpurun_out/r03_stamps2
for c in 3121 3221 3211 3220 3111; do
  JQC_EXTRA_DEFS="-DORED=1 -DPAROOT=1" timeout 300 python tools/stamps_profile.py $c 0112-elongated-nitrogenous >> gpurun_out/r03_stamps2/stamps_ored.txt 2>> gpurun_out/r03_stamps2/err.txt
done
JQC_JK_ALGO=v3361 JQC_EXTRA_DEFS="-DORED=1 -DPAROOT=1" timeout 300 python tools/stamps_profile.py 2121 0112-elongated-nitrogenous >> gpurun_out/r03_stamps2/stamps_ored.txt 2>> gpurun_out/r03_stamps2/err.txt
cat gpurun_out/r03_stamps2/stamps_ored.txt
