#!/bin/bash
# round 3: per-class times of the FP32 class kernels under the current variants (every quartet through FP32), 112 atoms / def2-TZVPP
export HSA_ENABLE_IPC_MODE_LEGACY=0
export JQC_TUNE_FP32=1
export JQC_TUNE_ONLY=0x22,0x1022,0x2022,0x3022,0x32,0x1032,0x40121,0xc0121,0x40521,0xc0521,0x40921,0xc0921,0x40d21,0xc0d21,0x121,0x921,0x521
timeout 1500 python tools/autotune.py run 0112-elongated-nitrogenous 2>&1 | grep -v amdgpu | tail -20
