#!/bin/bash
# round 3, A/B 4: general owner reduction on the 21 lane = (ci, cj) row-lane classes, in their tuned variants
export HSA_ENABLE_IPC_MODE_LEGACY=0
ROW=2022,2121,2122,2221,2222,3022,3031,3032,3033,3122,3131,3132,3133,3222,3231,3232,3233,3322,3331,3332,3333
JQC_AB_TAG=r03_ab4 timeout 1500 python tools/dev_ab.py run $ROW ored="-DORED=1 -DPAROOT=1" oredn="-DORED=1" > gpurun_out/r03_ab4.txt 2>&1
tail -25 gpurun_out/r03_ab4.txt
