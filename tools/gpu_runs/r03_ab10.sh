#!/bin/bash
# round 3: Rys root and weight polynomials one after the other (RYS_SPLIT=1: 14 coefficients in flight instead of 28), all 65 classes
export HSA_ENABLE_IPC_MODE_LEGACY=0
JQC_AB_TAG=rsplit timeout 1500 python tools/dev_ab.py run all "base=" "rsplit=-DRYS_SPLIT=1" > gpurun_out/r03_ab10.txt 2>&1
head -3 gpurun_out/r03_ab10.txt
