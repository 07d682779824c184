#!/bin/bash
# round 3, A/B 1: owner reduction (ORED) and per-root phase A (PAROOT) on the CJR row-lane classes
export HSA_ENABLE_IPC_MODE_LEGACY=0
CJR=2211,2220,3111,3121,3130,3211,3220,3221,3230,3310,3311,3320,3321,3330
JQC_AB_TAG=r03_ored_paroot timeout 1500 python tools/dev_ab.py run $CJR base= ored=-DORED=1 par=-DPAROOT=1 both="-DORED=1 -DPAROOT=1" > gpurun_out/r03_ab1.txt 2>&1
tail -30 gpurun_out/r03_ab1.txt
