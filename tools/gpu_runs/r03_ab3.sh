#!/bin/bash
# round 3, A/B 3: the 21 non-CJR row-lane classes forced to the j-in-registers form with owner reduction (+ wave-local steps)
export HSA_ENABLE_IPC_MODE_LEGACY=0
ROW=2022,2121,2122,2221,2222,3022,3031,3032,3033,3122,3131,3132,3133,3222,3231,3232,3233,3322,3331,3332,3333
JQC_AB_TAG=r03_ab3_base timeout 900 python tools/dev_ab.py run $ROW base= > gpurun_out/r03_ab3.txt 2>&1
JQC_AB_ALGO=0x921 JQC_AB_TAG=r03_ab3_cjr timeout 900 python tools/dev_ab.py run $ROW cjr="-DORED=1 -DPAROOT=1" >> gpurun_out/r03_ab3.txt 2>&1
JQC_AB_ALGO=0xd21 JQC_AB_TAG=r03_ab3_cjrw timeout 900 python tools/dev_ab.py run $ROW cjrw="-DORED=1 -DPAROOT=1" >> gpurun_out/r03_ab3.txt 2>&1
tail -5 gpurun_out/r03_ab3.txt
