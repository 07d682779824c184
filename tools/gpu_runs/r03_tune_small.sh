#!/bin/bash
# round 3: small-launch table of the row-lane classes (benzene / def2-TZVPP) with the owner-reduction forms
export HSA_ENABLE_IPC_MODE_LEGACY=0
export JQC_TUNE_SET=ored JQC_TUNE_ALL=1 JQC_TUNE_REPS=6
export JQC_TUNE_CLASSES=2022,2121,2122,2211,2220,2221,2222,3022,3031,3032,3033,3111,3121,3122,3130,3131,3132,3133,3211,3220,3221,3222,3230,3231,3232,3233,3310,3311,3320,3321,3322,3330,3331,3332,3333
export JQC_ONLY_CLASS=$JQC_TUNE_CLASSES
timeout 1500 python tools/autotune.py run benzene > gpurun_out/r03_tune_small.txt 2>&1
tail -5 gpurun_out/r03_tune_small.txt
