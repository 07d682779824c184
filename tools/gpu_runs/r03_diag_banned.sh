#!/bin/bash
# round 3: do the banned row-lane builds (one wave per SIMD, double-buffered TRR) still fail with the current sources?
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python tools/diag_banned_builds.py run > gpurun_out/r03_diag_banned.txt 2>&1
cat gpurun_out/r03_diag_banned.txt
