#!/bin/bash
# round-3 opening diagnostics: stamps of the dominant row-lane kernels + PMC profile (serial launches)
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r03_diag
for c in 3121 2121 3221 3220 3111 3211; do
  timeout 300 python tools/stamps_profile.py $c 0112-elongated-nitrogenous >> gpurun_out/r03_diag/stamps.txt 2>> gpurun_out/r03_diag/stamps.err
done
timeout 900 bash tools/pmc_profile.sh r03_diag/pmc 0112-elongated-nitrogenous
