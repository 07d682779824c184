#!/bin/bash
# round 3: which perturbation of the build cures the wrong results of the double-buffered TRR schedule?
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python tools/diag_dbuf.py run > gpurun_out/r03_diag_dbuf.txt 2>&1
cat gpurun_out/r03_diag_dbuf.txt
