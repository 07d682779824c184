#!/bin/bash
# round 3, A/B 2: owner reduction v2 (unrolled owner pass, whole quartets per wave), with / without wave-local steps and per-root phase A; TRR slot padding
export HSA_ENABLE_IPC_MODE_LEGACY=0
CJR=2211,2220,3111,3121,3130,3211,3220,3221,3230,3310,3311,3320,3321,3330
JQC_AB_TAG=r03_ored_v2 timeout 2400 python tools/dev_ab.py run $CJR base= ored=-DORED=1 oredw="-DORED=1 -DWSYNC=1" oredp="-DORED=1 -DPAROOT=1" oredwp="-DORED=1 -DWSYNC=1 -DPAROOT=1" tpad2="-DTPAD=2 -DPAROOT=1" tpad4="-DTPAD=4 -DPAROOT=1" > gpurun_out/r03_ab2.txt 2>&1
tail -30 gpurun_out/r03_ab2.txt
