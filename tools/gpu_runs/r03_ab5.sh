#!/bin/bash
# round 3, A/B 5: the larger lane-per-quartet classes under the row-lane forms with owner reduction
export HSA_ENABLE_IPC_MODE_LEGACY=0
T1Q=2111,2110,3021,3120,3210,3110,2021,2120,2011,2210,3011,1111,3020,2020,3030,3200,2200,3300,1011
JQC_AB_TAG=r03_ab5_base timeout 900 python tools/dev_ab.py run $T1Q base= > gpurun_out/r03_ab5.txt 2>&1
JQC_AB_ALGO=0x921 JQC_AB_TAG=r03_ab5_cjr timeout 900 python tools/dev_ab.py run $T1Q cjr="-DORED=1 -DPAROOT=1" >> gpurun_out/r03_ab5.txt 2>&1
JQC_AB_ALGO=0xd21 JQC_AB_TAG=r03_ab5_cjrw timeout 900 python tools/dev_ab.py run $T1Q cjrw="-DORED=1 -DPAROOT=1" >> gpurun_out/r03_ab5.txt 2>&1
JQC_AB_ALGO=0x421 JQC_AB_TAG=r03_ab5_rlw timeout 900 python tools/dev_ab.py run $T1Q rlw="-DORED=1" >> gpurun_out/r03_ab5.txt 2>&1
grep sum gpurun_out/r03_ab5.txt
