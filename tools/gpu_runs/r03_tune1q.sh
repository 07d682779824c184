#!/bin/bash
# round 3: re-tune the lane-per-quartet classes (ket pairs per iteration x waves per SIMD) on the clustered tiles
export HSA_ENABLE_IPC_MODE_LEGACY=0
export JQC_TUNE_ONLY=0x22,0x32,0x42,0x1022,0x1032,0x2022,0x2032,0x3022,0x3032,0x1042 JQC_TUNE_ALL=1 JQC_TUNE_REPS=3
export JQC_TUNE_CLASSES=0000,1000,1010,1011,1100,1110,1111,2000,2010,2011,2020,2021,2100,2110,2111,2120,2200,2210,3000,3010,3011,3020,3021,3030,3100,3110,3120,3200,3210,3300
export JQC_ONLY_CLASS=$JQC_TUNE_CLASSES
timeout 1500 python tools/autotune.py run 0112-elongated-nitrogenous > gpurun_out/r03_tune1q.txt 2>&1
cp gpurun_out/autotune_0112-elongated-nitrogenous.json gpurun_out/r03_autotune_tile1q.json
tail -12 gpurun_out/r03_tune1q.txt
