#!/bin/bash
# round 3: where the lane-per-quartet class kernels spend a workgroup's cycles (112 atoms / def2-TZVPP)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r03_stamps_tile1q.txt; : > $O
for c in 2111 2110 3021 3120 3210 3110 2010 1010; do timeout 200 python tools/stamps_profile.py $c 0112-elongated-nitrogenous 2>&1 | grep -v amdgpu >> $O; done
cat $O
