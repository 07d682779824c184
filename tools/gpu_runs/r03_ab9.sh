#!/bin/bash
# round 3: phase A of the row-lane kernels in two parts (roots per (quartet, root) lane, recurrences per (quartet, root, axis) lane)
export HSA_ENABLE_IPC_MODE_LEGACY=0
JQC_AB_TAG=pasplit timeout 1500 python tools/dev_ab.py run rowlane "base=" "split=-DPASPLIT=1" > gpurun_out/r03_ab9.txt 2>&1
cat gpurun_out/r03_ab9.txt
