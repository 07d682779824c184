#!/bin/bash
# round 3, A/B 6: ket tile pairs without canonical quartets skipped before staging; + re-tune of the lane-per-quartet classes
export HSA_ENABLE_IPC_MODE_LEGACY=0
SL=0000,1010,1111,2020,2121,2222,3030,3131,3232,3333,1000,2110,3121,2010
JQC_AB_TAG=r03_skip timeout 1200 python tools/dev_ab.py run $SL noskip="-DSKIP_EMPTY=0" skip= > gpurun_out/r03_ab6.txt 2>&1; tail -18 gpurun_out/r03_ab6.txt
bash tools/gpu_runs/r03_tune1q.sh
